"""GPU parity of the batched leaf operators of include/pbrt_hip.h against the oracle: bit-exact for
indices / lobes and for arithmetic inside the numeric contract; a few ulp where ocml meets libm."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _rays(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


@pytest.mark.parametrize("scene,kw,lo,hi", [("cbox.xml", dict(res=8), -0.95, 0.95), ("simple.xml", dict(res=8, spp=1), -6, 6),
                                            ("testring.xml", dict(res=8), -0.1, 0.1), ("us_sphere_box.xml", {}, -0.14, 0.14),
                                            ("us_cone_box.xml", {}, -0.14, 0.14),                         # analytic cone
                                            ("us_cone_box.xml", dict(tessellate="true"), -0.14, 0.14)])   # 896 triangles
def test_ray_intersect_and_ray_test(mi, ob, scene, kw, lo, hi):
    sc = mi.load_file(scene_path(scene), **kw)
    n = 20000
    o, d = _rays(n, 1, lo, hi)
    tmax = np.where(np.arange(n) % 3 == 0, 0.5 * (hi - lo), np.inf).astype(np.float32)
    got = sc.ray_intersect(o, d, tmax)
    t, prim, u, v = ob.OracleScene.from_scene(sc).ray_intersect(o, d, tmax)
    assert np.array_equal(got["prim"], prim)                 # ids: bit-exact
    assert np.array_equal(got["t"], t) and np.array_equal(got["u"], u) and np.array_equal(got["v"], v)
    assert 0.05 < got["valid"].mean() <= 1.0
    occ = sc.ray_test(o, d, tmax)
    assert np.array_equal(occ, ob.OracleScene.from_scene(sc).ray_test(o, d, tmax))
    assert np.array_equal(occ, got["valid"])
    hit = got["valid"]
    assert np.allclose(np.linalg.norm(got["n"][hit], axis=1), 1, atol=1e-5)
    assert np.allclose(got["p"][hit], o[hit] + got["t"][hit, None] * d[hit], atol=2e-4 * (hi - lo))


def test_axis_aligned_rays_through_shared_vertices(mi, ob):
    """zero direction components and origins in box faces (0 * inf in a slab test): the probe axis meets the centre
    vertex of the cone's base fan, shared by 96 triangles -- found, and the same primitive as the oracle picks"""
    o = np.array([[0, 0, 0], [0, 0, 0], [0.001, 0, 0], [0, 0, 0.2], [0.15, 0, 0.1]], np.float32)
    d = np.array([[0, 0, 1], [0, 1, 0], [0, 0, 1], [0, 0, -1], [-1, 0, 0]], np.float32)
    for kw in (dict(tessellate="true"), {}):     # the mesh, and the analytic cone (base disc centre at (0, 0, 0.06))
        sc = mi.load_file(scene_path("us_cone_box.xml"), **kw)
        got = sc.ray_intersect(o, d)
        t, prim, u, v = ob.OracleScene.from_scene(sc).ray_intersect(o, d, np.full(5, np.inf, np.float32))
        assert np.array_equal(got["prim"], prim) and np.array_equal(got["t"], t)
        assert got["valid"].all() and got["t"][0] == pytest.approx(0.06, rel=1e-5)


def test_unit_cone_known_hits_and_normals(mi):
    """analytic cone (closed unit cone under to_world): hand-computed hits, outward normals on the lateral surface
    and the base disc, also under a mirrored, non-uniformly scaled to_world"""
    s = 1 / np.sqrt(2)
    sc = mi.load_dict({"type": "scene", "c": {"type": "cone"}})
    r = sc.ray_intersect([[0, 0, -1], [2, 0, 0.5], [0, 0, 0.5], [0.25, 0.25, 0.25], [2, 0, 2], [0, 3, 0.5]],
                         [[0, 0, 1], [-1, 0, 0], [1, 0, 0], [0, 0, -1], [-1, 0, 0], [0, 1, 0]])
    assert list(r["valid"]) == [True, True, True, True, False, False]
    assert np.allclose(r["t"][:4], [1.0, 1.5, 0.5, 0.25], rtol=2e-6)
    assert np.allclose(r["n"][:4], [[0, 0, -1], [s, 0, s], [s, 0, s], [0, 0, -1]], atol=1e-6)
    assert np.allclose(r["p"][:4], [[0, 0, 0], [0.5, 0, 0.5], [0.5, 0, 0.5], [0.25, 0.25, 0]], atol=1e-6)
    T = mi.ScalarTransform4f
    tw = T().translate([0.5, -1, 2]) @ T().rotate([1, 2, 3], 40) @ T().scale([-0.5, 2, 3])
    sc = mi.load_dict({"type": "scene", "c": {"type": "cone", "to_world": tw}})
    M = tw.matrix
    rng = np.random.default_rng(3)
    q = rng.uniform(-1, 1, (12000, 3)) * [1.5, 1.5, 1] + [0, 0, 0.5]             # object-space origins around the cone
    o = q @ M[:3, :3].T + M[:3, 3]
    d = rng.normal(size=(12000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = sc.ray_intersect(o, d)
    hit = r["valid"]
    assert 0.08 < hit.mean() < 0.9
    Wi = np.linalg.inv(M)
    po = r["p"][hit].astype(np.float64) @ Wi[:3, :3].T + Wi[:3, 3]                 # hit points back in object space
    on_base = np.abs(po[:, 2]) < 1e-5
    on_side = np.abs(np.hypot(po[:, 0], po[:, 1]) - (1 - po[:, 2])) < 1e-4
    assert np.all(on_base | on_side) and on_base.sum() > 50 and on_side.sum() > 200
    # outward: the object-space image of the normal (covariant: n_obj ~ M^T n_world) is -z on the base and has a
    # positive radial component on the lateral surface
    no = r["n"][hit].astype(np.float64) @ M[:3, :3]
    no /= np.linalg.norm(no, axis=1, keepdims=True)
    pure_base = on_base & ~on_side
    assert np.allclose(no[pure_base], [0, 0, -1], atol=1e-5)
    side = on_side & ~on_base & (po[:, 2] < 0.98)
    want = np.stack([po[side, 0], po[side, 1], 1 - po[side, 2]], axis=1)
    want /= np.linalg.norm(want, axis=1, keepdims=True)
    assert np.allclose(no[side], want, atol=2e-4)


def test_empty_and_single_batches(mi):
    sc = mi.load_file(scene_path("cbox.xml"), res=8)
    r = sc.ray_intersect(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert r["t"].shape == (0,)
    r = sc.ray_intersect([[0, 0, 4]], [[0, 0, -1]])
    assert r["valid"][0] and r["t"][0] == pytest.approx(5.0, rel=1e-6)      # camera axis hits the back wall at z = -1
    assert sc.flatten()["prims"]["shape"][r["prim"][0]] == [s.id() for s in sc.shapes()].index("back")
    r = sc.ray_intersect([[0, 0, 4]], [[0, 0, 1]])
    assert not r["valid"][0] and np.isinf(r["t"][0]) and r["prim"][0] == 0xFFFFFFFF


@pytest.mark.parametrize("kind", ["diffuse", "conductor", "dielectric", "ultra_ref", "ultra_intent"])
def test_bsdf_sample_and_eval(mi, ob, capi, kind):
    n = 30000
    rng = np.random.default_rng(7)
    wi = rng.normal(size=(n, 3))
    wi = (wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32)
    ng = rng.normal(size=(n, 3))
    ng = (ng / np.linalg.norm(ng, axis=1, keepdims=True)).astype(np.float32)
    s1 = rng.random(n, dtype=np.float32)
    s2 = rng.random((n, 2), dtype=np.float32)
    quirks = capi.USQ_REFERENCE
    if kind == "diffuse":
        b = mi.DiffuseBSDF(mi.Properties("diffuse", dict(reflectance=[0.2, 0.5, 0.8])))
    elif kind == "conductor":
        b = mi.ConductorBSDF(mi.Properties("conductor"))
    elif kind == "dielectric":
        b = mi.DielectricBSDF(mi.Properties("dielectric"))
    else:
        quirks = capi.USQ_REFERENCE if kind == "ultra_ref" else 0
        b = mi.UltraBSDF(mi.Properties("ultrasound_bsdf", dict(impedance=7.8, roughness=0.7, quirks=quirks)))
    si = mi.SurfaceInteraction3f(wi, ng, ng)
    bs, w = b.sample(mi.BSDFContext(), si, s1, s2)
    m = b._material()
    wo, pdf, wref, lobe = ob.bsdf_sample(m, quirks, wi, ng, ng, s1, s2)
    assert np.array_equal(bs.sampled_component, lobe)
    assert np.array_equal(bs.wo, wo) and np.array_equal(bs.pdf, pdf)
    if kind.startswith("ultra"):
        assert np.array_equal(w, wref[:, 0])
        assert b.eval_pdf(None, si, wo) == (0.0, 0.0)
    else:
        assert np.array_equal(w, wref)
        f, p = b.eval_pdf(None, si, wo)
        fr, pr = ob.bsdf_eval_pdf(m, wi, wo)
        assert np.array_equal(f, fr) and np.array_equal(p, pr)
    valid = lobe != 0xFFFFFFFF
    assert valid.mean() > 0.4
    if kind.startswith("ultra"):
        # with the tangent of the interaction's shading frame (Mitsuba builds sh_frame from dp_du): only bs.wo moves
        sh_s = rng.normal(size=(n, 3)).astype(np.float32)
        bs2, w2 = b.sample(mi.BSDFContext(), mi.SurfaceInteraction3f(wi, ng, ng, sh_s=sh_s), s1, s2)
        wo2, pdf2, wref2, lobe2 = ob.bsdf_sample(m, quirks, wi, ng, ng, s1, s2, sh_s=sh_s)
        assert np.array_equal(bs2.wo, wo2) and np.array_equal(bs2.pdf, pdf) and np.array_equal(w2, w)
        assert not np.array_equal(wo2, wo) and np.allclose(np.linalg.norm(wo2, axis=1), np.linalg.norm(wo, axis=1), rtol=1e-4)


def test_reference_style_scalar_sample2(mi, ob, capi):
    """The reference calls bsdf.sample(ctx, si, Float(s1), Float(s2)) with scalar samples (CustomIntegrator.py:338)."""
    b = mi.UltraBSDF(mi.Properties("ultrasound_bsdf", dict(impedance=7.8, roughness=0.9)))
    si = mi.SurfaceInteraction3f([[0.1, -0.2, 0.97]], [[0, 0, 1]], [[0, 0, 1]])
    bs, a_resp = b.sample(mi.BSDFContext(), si, 0.3, 0.6)
    wo, pdf, w, lobe = ob.bsdf_sample(b._material(), capi.USQ_REFERENCE, si.wi, [0, 0, 1], [0, 0, 1], 0.3, [[0.6, 0.6]])
    assert a_resp.shape == (1,) and a_resp[0] == w[0, 0] and np.array_equal(bs.wo, wo) and bs.eta[0] == 1.0


def test_emitter_sample_direction(mi, ob):
    for scene, kw in (("cbox.xml", dict(res=8)), ("simple.xml", dict(res=8, spp=1))):
        sc = mi.load_file(scene_path(scene), **kw)
        rng = np.random.default_rng(2)
        p = rng.uniform(-0.9, 0.9, (10000, 3)).astype(np.float32)
        u = rng.random((10000, 4), dtype=np.float32)
        got = sc.sample_emitter_direction(p, u)
        ref = ob.OracleScene.from_scene(sc).sample_emitter_direction(p, u)
        for k in ("d", "dist", "pdf", "weight", "p", "emitter"):
            assert np.array_equal(got[k], ref[k]), k
        assert (got["pdf"] > 0).mean() > 0.9
    # cbox luminaire: points lie on it, weights are radiance / pdf
    sc = mi.load_file(scene_path("cbox.xml"), res=8)
    got = sc.sample_emitter_direction(np.zeros((1000, 3), np.float32), np.random.default_rng(0).random((1000, 4), dtype=np.float32))
    assert np.allclose(got["p"][:, 1], 0.99, atol=1e-6) and np.all(np.abs(got["p"][:, [0, 2]]) <= 0.25 + 1e-6)
    assert np.allclose(got["weight"][:, 0] * got["pdf"], 1.0, rtol=1e-5)


def test_perspective_sensor_sample_ray(mi, ob):
    sc = mi.load_file(scene_path("cbox.xml"), res=64)
    sens = sc.sensors()[0]
    pos = np.random.default_rng(0).random((5000, 2), dtype=np.float32)
    ray, w = sens.sample_ray(0.0, 0.0, pos, None)
    o, d, tmax = ob.sensor_sample_ray(sens.camera(), pos)
    assert np.array_equal(ray["o"], o) and np.array_equal(ray["d"], d) and np.array_equal(ray["maxt"], tmax)
    c, _ = sens.sample_ray(0, 0, [[0.5, 0.5]], None)
    assert np.allclose(c["d"][0], [0, 0, -1], atol=1e-6) and np.allclose(c["o"][0], [0, 0, 4 - 0.001], atol=1e-6)
    l, _ = sens.sample_ray(0, 0, [[0.0, 0.5]], None)
    assert l["d"][0, 0] < 0          # film-left looks towards world -x (green wall)


def test_ultra_sensor_and_custom_emitter_sample_ray(mi, ob):
    rng = np.random.default_rng(5)
    n = 8000
    pos, ap = rng.random((n, 2), dtype=np.float32), rng.random((n, 2), dtype=np.float32)
    t, wl = (rng.random(n, dtype=np.float32) * 1e-6), rng.random(n, dtype=np.float32)
    for props in (dict(num_elements_lateral=64, pitch=3e-4), dict(num_elements_lateral=32, pitch=3e-4, radius=0.04)):
        s = mi.UltraSensor(mi.Properties("ultrasound_sensor", dict(props, to_world=mi.ScalarTransform4f().look_at(
            [0, 0, 0], [0.1, 0, 1], [0, 1, 0]))))
        for hemi in (True, False):
            ray, w = s.sample_ray(t, wl, pos, ap, use_hemisphere_warp=hemi)
            o, d, wr = ob.us_sensor_sample_ray(s._desc(), int(hemi), t, wl, pos, ap)
            assert np.allclose(ray["o"], o, atol=1e-7) and np.allclose(ray["d"], d, atol=2e-6) and np.allclose(w, wr, atol=2e-5)
            if hemi and "radius" not in props:
                assert np.array_equal(ray["d"], d)      # concentric warp only: inside the numeric contract
    for props in (dict(), dict(radius=0.05, opening_angle=60.0, number_of_elements=48, number_of_rays_per_element=4)):
        e = mi.CustomEmitter(mi.Properties("ultrasound_emitter", props))
        s1, s3 = rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32)
        ray, w = e.sample_ray(t, s1, pos, s3)
        o, d, rt, wr, pdf = ob.us_emitter_sample_ray(e._desc(), t, s1, pos, s3)
        assert np.allclose(ray["o"], o, atol=1e-7) and np.allclose(ray["d"], d, atol=2e-6)
        assert np.allclose(ray["time"], rt, atol=1e-9) and np.allclose(w, wr, atol=2e-6)
        ps, pdf_pos = e.sample_position(t, (s1, pos))
        assert np.allclose(ps["p"], o, atol=1e-7) and np.allclose(pdf_pos, pdf, rtol=1e-6)
        assert pdf_pos[0] == pytest.approx(1.0 / (e.number_of_elements * e.element_width * e.element_height), rel=1e-5)


def test_put_data_k6_on_device(mi, known):
    k = known["K6_put_data"]
    s = mi.plugins.CustomSensor(mi.Properties("custom_sensor", dict(number_of_elements=k["number_of_elements"], pitch=k["pitch"],
                                                                   sample_rate=k["sample_rate"], time_samples=k["time_samples"])))
    rays = k["rays"]
    s.put_data(dict(o=[[r["x"], 0, 0] for r in rays], d=[r["d"] for r in rays], time=[r["time"] for r in rays]),
               [r["amplitude"] for r in rays])
    buf = s.channel_data()
    nz = {(int(i), int(j)): float(buf[i, j]) for i, j in np.argwhere(buf != 0)}
    want = {(e["element"], e["sample"]): e["value"] for e in k["nonzero"]}
    assert nz.keys() == want.keys() and all(abs(nz[q] - want[q]) < 1e-6 for q in want)
    s.put_data(dict(o=[[-2.0, 0, 0]], d=[[0, 0, -1]], time=[1.0]), [0.5])     # accumulates
    assert s.channel_data()[0, 10] == pytest.approx(1.5)
    s.clear()
    assert not s.channel_data().any()
