"""Shared by tests/test_pinned_transcription.py (oracle, CPU) and tests/test_gpu_pinned.py (HIP): load the K9 / K10
fixtures made by tests/golden/make_pinned.py from the float64 transcription of the reference (ref_transcription.py),
rebuild their scenes through the package's dict API, and compare an implementation's output with them.

Tolerances (f32 implementation against an f64 restatement in another operation order):
  * single-bounce BSDF records: pdf, amplitude, wo to 1e-4 relative;
  * echo bins: the carrier sin(2 pi f t) is evaluated at phase ~ 2.6e3 rad, where f32 keeps ~ 2e-4 rad -> |dp| <= 2e-3 of
    the bin's summed |envelope|; the envelopes themselves (PBRT_USQ_NO_CARRIER deposits them) to 5e-4 relative;
  * only bins / samples whose smallest decision margin (distance of a random draw from the branch it decides, of a hit from
    a shape's edge, of a clamped denominator from its clamp) is >= 1e-2 are compared by value -- closer than that an f32
    and an f64 evaluation may legitimately take different branches; the fixture says which."""
import json
import os

import numpy as np

from conftest import GOLDEN

SAFE = 1e-2


def rng4(a, b, c, seed):
    """the library's counter-based generator pcg4d(a, b, c, seed) -> four 24-bit uniforms (csrc/device_math.h rng4, oracle/omath.h),
    restated on Python integers: what a test needs to build the inputs of a leaf operator without importing the generators of the
    pinned fixtures (tests/golden/ref_transcription.py stays in the build container)"""
    m = 0xFFFFFFFF
    x, y, z, w = [(v * 1664525 + 1013904223) & m for v in (a, b, c, seed)]
    x = (x + y * w) & m
    y = (y + z * x) & m
    z = (z + x * y) & m
    w = (w + y * z) & m
    x ^= x >> 16
    y ^= y >> 16
    z ^= z >> 16
    w ^= w >> 16
    x = (x + y * w) & m
    y = (y + z * x) & m
    z = (z + x * y) & m
    w = (w + y * z) & m
    return [(v >> 8) / 16777216.0 for v in (x, y, z, w)]


def load_k12():
    z = np.load(os.path.join(GOLDEN, "k12_emitter_sensor.npz"))
    return z, json.loads(str(z["meta"]))


def k12_emitter(mi, props):
    return mi.CustomEmitter(mi.Properties("ultrasound_emitter", dict(props)))


def k12_sensor(mi, props, look_at):
    p = {k: (float("inf") if v is None else v) for k, v in props.items()}
    return mi.UltraSensor(mi.Properties("ultrasound_sensor", dict(p, to_world=mi.ScalarTransform4f().look_at(*look_at))))


def load_k9(name):
    z = np.load(os.path.join(GOLDEN, f"k9_us_{name}.npz"))
    return z, json.loads(str(z["meta"]))


def k9_scene(mi, meta):
    """the fixture's scene through load_dict (matrices as written in the fixture, not the XML loader)"""
    P = meta["params"]
    T = mi.ScalarTransform4f
    d = {"type": "scene",
         "integrator": {"type": "ultrasound_integrator", "max_depth": P["max_depth"], "sampling_rate": P["fs"], "frequency": P["frequency"],
                        "sound_speed": P["sound_speed"], "attenuation": P["attenuation"], "main_beam_angle": P["main_beam_angle"],
                        "cutoff_angle": P["cutoff_angle"], "n_elements": P["n_elements"], "pitch": P["pitch"],
                        "time_samples": P["time_samples"], "angles": np.asarray(P["angles_deg"], np.float32),
                        "paths_per_ray": meta["paths_per_ray"], "seed": meta["seed"]},
         "sensor": {"type": "ultrasound_sensor", "to_world": T().look_at(*meta["look_at"])}}
    if "emitter" in meta:   # every path draws its primary ray from the scene's CustomEmitter (PBRT_US_PRIMARY_EMITTER)
        d["integrator"]["primary_rays"] = "emitter"
        d["emitter"] = dict(meta["emitter"], type="ultrasound_emitter")
    for i, s in enumerate(meta["shapes"]):
        bs = {"type": "ultrasound_bsdf", "impedance": s["impedance"], "roughness": s["roughness"]}
        if s["type"] == "sphere":
            d[f"shape{i}"] = {"type": "sphere", "center": s["center"], "radius": s["radius"], "bsdf": bs}
        elif s["type"] == "obj":
            from conftest import scene_path
            d[f"shape{i}"] = {"type": "obj", "filename": scene_path(s["filename"]), "to_world": T(np.asarray(s["to_world"])), "bsdf": bs}
        else:
            d[f"shape{i}"] = {"type": "rectangle", "to_world": T(np.asarray(s["to_world"])), "bsdf": bs}
    return mi.load_dict(d)


def check_k9_bins(z, meta, buf, carrier=True):
    """buf: channel buffer [n_angles, n_elements, T] normalised by paths_per_ray (as the library returns it)"""
    got = np.asarray(buf, np.float64) * meta["paths_per_ray"]
    idx, margin = z["bin_index"], z["bin_margin"]
    safe = margin >= SAFE
    assert safe.sum() >= 0.7 * len(idx) and safe.sum() >= 500
    g = got[idx[:, 0], idx[:, 1], idx[:, 2]]
    # arrival: every well-conditioned echo lands in the bin the transcription says, and nothing lands elsewhere
    # except where an ill-conditioned decision may have moved it
    assert np.all(g[safe & (z["bin_envelope_abs"] > 0)] != 0)        # (an echo outside the cut-off cone deposits an exact 0)
    mask = np.zeros(got.shape, bool)
    mask[idx[:, 0], idx[:, 1], idx[:, 2]] = True
    stray = np.count_nonzero(got[~mask])
    assert stray <= (~safe).sum()
    env_abs = np.maximum(z["bin_envelope_abs"], 1e-300)
    if carrier:
        err = np.abs(g - z["bin_pressure"]) / env_abs
        assert err[safe].max() <= 2e-3, err[safe].max()
    else:
        err = np.abs(g - z["bin_envelope"]) / env_abs
        assert err[safe].max() <= 5e-4, err[safe].max()
    return float(err[safe].max())


def check_k9_records(z, meta, sample_fn):
    """sample_fn(impedance, roughness, wi, n, sh_s, s1, s2[, sh_n]) -> (wo [k,3], pdf [k], amp [k], lobe [k]) for one material
    (sh_n: si.sh_frame.n where it differs from the geometric normal -- meshes with vertex normals; fixtures made since round 4)"""
    shapes = meta["shapes"]
    worst = 0.0
    for si in sorted(set(z["rec_shape"].tolist())):
        sel = z["rec_shape"] == si
        extra = {}
        if "rec_sh_n" in z.files and not np.array_equal(z["rec_sh_n"][sel], z["rec_n"][sel]):
            extra["sh_n"] = z["rec_sh_n"][sel].astype(np.float32)
        wo, pdf, amp, lobe = sample_fn(shapes[si]["impedance"], shapes[si]["roughness"], z["rec_wi"][sel].astype(np.float32),
                                       z["rec_n"][sel].astype(np.float32), z["rec_sh_s"][sel].astype(np.float32),
                                       z["rec_s1"][sel].astype(np.float32), z["rec_s2"][sel].astype(np.float32), **extra)
        assert np.array_equal(lobe == 0, z["rec_reflect"][sel])
        assert np.allclose(pdf, z["rec_pdf"][sel], rtol=1e-4, atol=0)
        assert np.allclose(amp, z["rec_a_resp"][sel], rtol=1e-4, atol=1e-7)
        scale = np.linalg.norm(z["rec_wo"][sel], axis=1, keepdims=True)
        assert np.all(np.abs(wo - z["rec_wo"][sel]) <= 1e-4 * scale)
        # the integrator's next direction is normalize(si.to_world(bs.wo)) (CustomIntegrator.py:358-359)
        worst = max(worst, float(np.max(np.abs(pdf / z["rec_pdf"][sel] - 1))))
    # what the records cover: plate = reflection / transmission / total internal reflection at the first bounce;
    # sphere_box (roughness 0.9: every well-conditioned facet is past the 8.85 degree TIR angle) adds second bounces
    if meta["scene"] == "plate":
        assert z["rec_reflect"].sum() >= 8 and (~z["rec_reflect"]).sum() >= 4 and z["rec_tir"].sum() >= 4
    elif meta["scene"] == "testring":
        # the mesh phantom: second bounces inside the ring's wall (none of them seen by an element), shading normals on the cylinders
        assert (z["rec_depth"] >= 1).sum() >= 8 and z["rec_tir"].sum() >= 8 and "rec_sh_n" in z.files
    elif meta["scene"] == "plate_box":
        # MitsubaScenes/Plate_Box.xml: every path ends at its first bounce -- the plate is tilted by 45 degrees, so whatever the
        # facet, the new direction leaves the 30-degree cut-off cone about the transducer normal (CustomIntegrator.py:368-372) and
        # the five walls are never reached: 960 first-bounce echoes, none later (the fixture records that, it was not expected)
        assert list(meta["depth_histogram"]) == ["0"] and z["rec_tir"].sum() >= 8
    else:
        assert (z["rec_depth"] >= 1).sum() >= 8 and z["rec_tir"].sum() >= 8
    if meta["scene"] == "two_plates":   # the scene whose second-bounce echoes are deposited
        assert int(meta["deposited_by_depth"]["1"]) >= 100
    return worst


def load_k10():
    z = np.load(os.path.join(GOLDEN, "k10_cbox_paths.npz"))
    return z, json.loads(str(z["meta"]))


def check_k10(z, meta, render_sample):
    """render_sample(s) -> [res, res, 3] radiance of sample index s of every pixel (box filter, 1 spp)"""
    worst = 0.0
    for s in range(meta["samples"]):
        got = np.asarray(render_sample(s), np.float64)
        want, margin = z["radiance"][s], z["margin"][s]
        safe = margin >= 1e-3
        assert safe.mean() >= 0.9
        err = np.abs(got - want).max(axis=2) / np.maximum(want.max(axis=2), 1e-2)
        assert err[safe].max() <= 2e-4, (err[safe].max(), np.argwhere(err * safe == err[safe].max()))
        # the ill-conditioned ones may take another branch, but not many of them do
        assert (err[~safe] > 1e-3).sum() <= 3
        worst = max(worst, float(err[safe].max()))
        assert (want[safe].sum(axis=1) > 0).mean() > 0.5
    return worst


def load_k11():
    z = np.load(os.path.join(GOLDEN, "k11_meshes.npz"))
    return z, json.loads(str(z["meta"]))


def check_k11(want, render_sample, n_samples):
    """triangle meshes: per-sample radiance against the float64 restatement.  No decision margins here (a shadow ray that
    grazes a silhouette edge of a 2 256-triangle mesh has no cheap one): instead at least 99.5 % of the samples must agree
    to 2e-4 of max(radiance, 0.01), the lit / unlit pattern of the rest may differ, and the means must agree to 1e-3"""
    worst_frac = 1.0
    for s in range(n_samples):
        got = np.asarray(render_sample(s), np.float64)
        w = np.asarray(want[s], np.float64)
        err = np.abs(got - w).max(axis=2) / np.maximum(w.max(axis=2), 1e-2)
        frac = float((err <= 2e-4).mean())
        worst_frac = min(worst_frac, frac)
        assert frac >= 0.995, frac
        assert abs(got.mean() - w.mean()) <= 1e-3 * w.mean()
        assert (w.sum(axis=2) > 0).mean() > 0.1
    return worst_frac
