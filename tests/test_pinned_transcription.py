"""The CPU oracle against K9 / K10: fixtures produced by an INDEPENDENT float64 restatement of the reference's
hot loop (tests/golden/ref_transcription.py: written from CustomIntegrator.py:262-376 and CustomBSDF.py:30-175, and from
Mitsuba's published semantics for the radiance path -- not from oracle/oracle.cpp, with which it shares no source).
A sign, a frame, a pdf multiplied instead of divided, a wrong MIS weight in oracle.cpp's `ultra_core` / `path_radiance`
fails here, where the oracle-vs-HIP tests (same leaf arithmetic on both sides) cannot see it."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, oracle_render, scene_path
from pinned_util import SAFE, check_k10, check_k11, check_k9_bins, check_k9_records, k9_scene, load_k10, load_k11, load_k9

sys.path.insert(0, GOLDEN)


@pytest.mark.parametrize("name", ["plate", "sphere_box", "two_plates", "plate_box", "testring", "sphere_box_emitter"])
def test_k9_fixture_is_what_the_transcription_produces(name):
    """the committed fixture equals a fresh run of the transcription (first rays of every angle)"""
    import make_pinned as mp
    import ref_transcription as rt
    z, meta = load_k9(name)
    S = mp.US_SCENES[name]
    assert meta["params"] == S["params"] and meta["seed"] == S["seed"]
    shapes, T = mp.build_shapes(S["shapes"]), rt.look_at(*S["look_at"])
    have = {tuple(k): (p, e) for k, p, e in zip(z["bin_index"].tolist(), z["bin_pressure"], z["bin_envelope"])}
    NE, n = S["params"]["n_elements"], 0
    acc = {}
    for a in range(len(S["params"]["angles_deg"])):
        for e in range(NE):
            ray = a * NE + e
            for k in range(S["ppr"]):
                for r in rt.us_trace_single_ray(shapes, T, S["params"], a, e, lambda dep: rt.rng4(ray, k, dep, S["seed"]),
                                                primary=mp.emitter_primary(S, a, e, ray, k)):
                    if r["deposited"]:
                        key = (a, r["recv"], r["t_idx"])
                        acc[key] = acc.get(key, 0.0) + r["pressure"]
                        n += 1
    assert set(acc) == set(have)
    assert all(abs(acc[k] - have[k][0]) <= 1e-12 * max(1.0, abs(acc[k])) for k in acc)


def test_rng_of_the_transcription_is_the_librarys(ob, capi):
    """draws are inputs of a fixture; they come from the same counter-based generator on both sides (pcg4d): the
    diffuse BSDF passes its 2-D sample through the concentric disk, which pins u.y, u.z of a block; the camera jitter
    of a render pins block 0 -- checked end to end by K10.  Here: the integer hash itself."""
    import ref_transcription as rt
    assert rt.pcg4d(0, 0, 0, 0) != rt.pcg4d(0, 0, 0, 1)
    u = np.array([rt.rng4(a, b, c, 9) for a, b, c in [(0, 0, 0), (1, 2, 3), (4095, 77, 11), (2 ** 31, 5, 1)]])
    assert np.all((u >= 0) & (u < 1)) and len(np.unique(u)) == u.size


@pytest.mark.parametrize("name", ["plate", "sphere_box", "two_plates", "plate_box", "testring", "sphere_box_emitter"])
def test_k9_oracle_single_bounce_records(ob, capi, name):
    z, meta = load_k9(name)

    def sample(imp, rough, wi, n, sh_s, s1, s2, sh_n=None):
        m = capi.make_material(capi.MAT_ULTRA, [imp, rough, 1.2])
        wo, pdf, w, lobe = ob.bsdf_sample(m, capi.USQ_REFERENCE, wi, n, n if sh_n is None else sh_n, s1, np.stack([s2, s2], axis=1), sh_s=sh_s)
        return wo, pdf, w[:, 0], lobe
    check_k9_records(z, meta, sample)


@pytest.mark.parametrize("name", ["plate", "sphere_box", "two_plates", "plate_box", "testring", "sphere_box_emitter"])
def test_k9_oracle_echo_values(mi, ob, capi, name):
    """every echo of every path: arrival bin, pressure (CustomIntegrator.py:340-354) and, with the carrier off, the
    envelope atten * amp * w_i * w_o alone; in two_plates the second-bounce echoes reach the receive elements (162 of
    them), so the continuation direction (:358-359), the roulette division (:364-367) and the accumulated time of flight
    (:316) are in deposited values (plate: paths end after one bounce; sphere_box: they go on inside the sphere, unseen)"""
    z, meta = load_k9(name)
    sc = k9_scene(mi, meta)
    ui = sc.integrator()
    osc = ob.OracleScene.from_scene(sc)
    buf, _ = osc.us_acquire(ui.us_params(sc), meta["seed"], meta["paths_per_ray"])
    check_k9_bins(z, meta, buf, carrier=True)
    buf, _ = osc.us_acquire(ui.us_params(sc, ui.quirks | capi.USQ_NO_CARRIER), meta["seed"], meta["paths_per_ray"])
    check_k9_bins(z, meta, buf, carrier=False)
    if name == "two_plates":
        assert int(meta["deposited_by_depth"]["1"]) >= 100


def test_k10_oracle_path_values(mi, ob):
    """per-sample radiance of the Cornell box (6 bounces: emission + MIS, emitter sampling + shadow ray, diffuse /
    mirror / glass sampling, roulette) against the float64 Mitsuba-`path` restatement"""
    z, meta = load_k10()
    sc = mi.load_file(scene_path("cbox.xml"), res=meta["res"], spp=1, max_depth=meta["max_depth"], rfilter="box")
    worst = check_k10(z, meta, lambda s: oracle_render(ob, sc, meta["seed"], 1, sample_offset=s)[0])
    assert worst < 2e-4


def test_k11_oracle_simple_xml_and_shading_normals(mi, ob, tmp_path):
    """K11: BASELINE config 1 (scenes/simple.xml: teapot.ply, `direct`, two point emitters, box filter, default 50 mm lens) sample
    by sample, and a 16-triangle ball with vertex normals (interpolated shading normal), against the float64 restatement"""
    from mesh_util import write_uv_sphere_obj
    z, meta = load_k11()
    m = meta["simple"]
    sc = mi.load_file(scene_path("simple.xml"), res=m["res"], spp=1)
    assert sc.sensors()[0].x_fov == pytest.approx(m["x_fov"], abs=1e-9)
    check_k11(z["simple"], lambda s: oracle_render(ob, sc, m["seed"], 1, sample_offset=s)[0], m["samples"])
    b = meta["ball"]
    write_uv_sphere_obj(str(tmp_path / "ball.obj"), n_lat=b["n_lat"], n_lon=b["n_lon"], normals=True)
    check_k11(z["ball"], lambda s: oracle_render(ob, _ball_scene(mi, tmp_path, b), b["seed"], 1, sample_offset=s)[0], b["samples"])


def _ball_scene(mi, tmp_path, b):
    T = mi.ScalarTransform4f
    return mi.load_dict({
        "type": "scene", "integrator": {"type": "path", "max_depth": b["max_depth"]},
        "sensor": {"type": "perspective", "fov": 30, "near_clip": 0.1, "far_clip": 50,
                   "to_world": T().look_at([0, 0, 5], [0, 0, 0], [0, 1, 0]),
                   "film": {"type": "hdrfilm", "width": b["res"], "height": b["res"], "rfilter": {"type": "box"}},
                   "sampler": {"type": "independent", "sample_count": 1}},
        "ball": {"type": "obj", "filename": str(tmp_path / "ball.obj"),
                 "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.7, 0.6]}}},
        "bulb": {"type": "point", "position": [3, 4, 6], "intensity": {"type": "rgb", "value": [60, 60, 60]}}})


def test_k9_oracle_drjit_variant(mi, ob, capi):
    """simulate_acquisition, the dr.while_loop variant (CustomIntegrator.py:60-232): draws frozen at trace time (every bounce of a
    ray reuses its first draws), tof of the current segment only, clamped time bins, signed strict roulette"""
    z, meta = load_k9("two_plates_drjit")
    assert meta["variant"] == "drjit"
    sc = k9_scene(mi, meta)
    ui = sc.integrator()
    osc = ob.OracleScene.from_scene(sc)
    q = ui.quirks | capi.USQ_DRJIT_VARIANT
    buf, _ = osc.us_acquire(ui.us_params(sc, q), meta["seed"], meta["paths_per_ray"])
    check_k9_bins(z, meta, buf, carrier=True)
    buf, _ = osc.us_acquire(ui.us_params(sc, q | capi.USQ_NO_CARRIER), meta["seed"], meta["paths_per_ray"])
    check_k9_bins(z, meta, buf, carrier=False)
    # it is a different estimator from the scalar variant's: second-bounce echoes land elsewhere (tof is not accumulated)
    zs, _ = load_k9("two_plates")
    assert not np.array_equal(z["bin_index"], zs["bin_index"]) and int(meta["deposited_by_depth"]["1"]) >= 100


def test_k12_emitter_and_sensor_oracle(mi, ob):
    """rows a13 / a14: the oracle's CustomEmitter.sample_ray / sample_position and UltraSensor.sample_ray against the float64
    transcription of CustomEmmitter.py:30-107 (linear array and the convex branch :41-47: steering delay -x sin(psi) / c, weight
    max(0, d.n) / N_rays) and of SURVEY App. C -- code that shares no source text with oracle.cpp.  1e-5."""
    from pinned_util import load_k12, k12_emitter, k12_sensor
    z, meta = load_k12()
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    t, s1, s2, s3, wl, pos, ap = (f32(z[k]) for k in ("time", "s1", "s2", "s3", "wl", "pos", "ap"))
    for name, P in meta["emitters"].items():
        e = k12_emitter(mi, P)
        o, d, rt_, w, pdf = ob.us_emitter_sample_ray(e._desc(), t, s1, s2, s3)
        want = z[f"emitter_{name}"]
        assert np.allclose(o, want[:, 0:3], atol=1e-7) and np.allclose(d, want[:, 3:6], atol=1e-5)
        assert np.allclose(rt_, want[:, 6], atol=1e-10, rtol=1e-5) and np.allclose(w, want[:, 7], atol=1e-5 * want[:, 7].max())
        assert np.allclose(pdf, want[:, 8], rtol=1e-5)
    for name, P in meta["sensors"].items():
        sn = k12_sensor(mi, P, meta["look_at"])
        for hemi in (1, 0):
            o, d, w = ob.us_sensor_sample_ray(sn._desc(), hemi, t, wl, pos, ap)
            want = z[f"sensor_{name}_{hemi}"]
            assert np.allclose(o, want[:, 0:3], atol=1e-7) and np.allclose(d, want[:, 3:6], atol=1e-5) and np.allclose(w, want[:, 6], atol=2e-5)
