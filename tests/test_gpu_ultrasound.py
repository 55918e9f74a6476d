"""GPU parity, ultrasound mode: UltraIntegrator.simulate_acquisition_parallel (CustomIntegrator.py:235-405)
on the HIP wavefront path vs the CPU oracle.

Tolerance: the per-path arithmetic differs from the oracle only through sinf/cosf/expf/acosf (ocml vs libm,
a few ulp) and the channel buffer is a sum of f32 atomics in arbitrary order (the oracle sums in f64 in path
order), so parity is stated as relative L2 <= 1e-3 of the whole channel buffer (north star's tolerance) and,
per bin, |a-b| <= 1e-4 * max|ref| ; the set of non-zero bins must be identical."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path

pytestmark = pytest.mark.gpu
TOL_REL_L2 = 1e-3


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-300))


def check(buf, ref):
    assert buf.shape == ref.shape and np.isfinite(buf).all()
    assert rel_l2(buf, ref) <= TOL_REL_L2
    assert np.abs(buf - ref).max() <= 1e-4 * np.abs(ref).max()
    assert np.array_equal(buf != 0, ref != 0)


@pytest.mark.parametrize("scene,ppr,seed,kw", [("us_plate.xml", 64, 0, {}), ("us_plate.xml", 500, 3, {}), ("us_sphere_box.xml", 200, 1, {}),
                                               ("us_cone_box.xml", 100, 2, {}),                          # analytic cone, brute force
                                               ("us_cone_box.xml", 100, 2, dict(tessellate="true")),     # 901 triangles, BVH
                                               # the other phantoms of the reference's MitsubaScenes/ (author-intent transforms)
                                               ("us_sphere_floating.xml", 150, 4, {}), ("us_plane_floating.xml", 150, 5, {}),
                                               ("us_plate_box.xml", 150, 6, {}), ("us_cone_floating.xml", 150, 7, {}),
                                               ("us_cone_floating.xml", 60, 8, dict(tessellate="true")),
                                               # the ring of BASELINE config 4 as phantom (1157 primitives: k_trace + k_us_shade): with the
                                               # first-bounce tables, and with fewer paths per ray than elements (primary rays traced)
                                               ("us_testring.xml", 100, 9, {}), ("us_testring.xml", 24, 10, {}),
                                               # BASELINE config 3 "with CustomBSDF + CustomEmmitter": every path's primary ray from
                                               # CustomEmitter.sample_ray (k_us_bounce<true, ., EMIT>; no first-bounce tables)
                                               ("us_sphere_box.xml", 300, 11, dict(primary_rays="emitter")),
                                               ("us_sphere_box.xml", 16, 12, dict(primary_rays="emitter"))])
def test_acquisition_matches_oracle(mi, ob, scene, ppr, seed, kw):
    sc = mi.load_file(scene_path(scene), paths_per_ray=ppr, seed=seed, **kw)
    ui = sc.integrator()
    assert ui.simulate_acquisition_parallel(sc) is True                # CustomIntegrator.py:405
    ref, tx = ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc), seed, ppr)
    assert ui.channel_buf.shape == (ui.n_angles, ui.n_elements, ui.time_samples)
    check(ui.channel_buf, ref)
    assert np.array_equal(ui.transmission_delays_buf, tx)
    st = mi.default_context().stats()
    assert st["samples"] == ui.n_angles * ui.n_elements * ppr and ui.ray_count == st["segments"] > 0


@pytest.mark.parametrize("scene,ppr,name", [("us_plate.xml", 32, "us_plate_ppr32_seed0.npz"), ("us_cone_box.xml", 8, "us_cone_box_ppr8_seed0.npz")])
def test_golden_channel_buffer(mi, scene, ppr, name):
    g = np.load(os.path.join(GOLDEN, name))
    sc = mi.load_file(scene_path(scene), paths_per_ray=ppr, seed=0)
    ui = sc.integrator()
    ui.simulate_acquisition_parallel(sc)
    ref = np.zeros_like(ui.channel_buf)
    ref[tuple(g["index"].T)] = g["value"]
    check(ui.channel_buf, ref)
    assert np.array_equal(ui.transmission_delays_buf, g["tx"])


def test_drjit_variant_semantics(mi, ob, capi):
    """simulate_acquisition (CustomIntegrator.py:60-232): draws frozen at trace time, clamped time bins, no tof accumulation,
    signed roulette; flat buffer."""
    sc = mi.load_file(scene_path("us_plate.xml"), paths_per_ray=128, seed=2)
    ui = sc.integrator()
    ui.simulate_acquisition(sc)
    assert ui.channel_buf.shape == (ui.n_angles * ui.n_elements * ui.time_samples,)
    q = ui.quirks | capi.USQ_DRJIT_VARIANT
    ref, _ = ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc, q), 2, 128)
    check(ui.channel_buf.reshape(ref.shape), ref)


@pytest.mark.parametrize("quirks", ["intent", "mixed"])
def test_quirk_switches(mi, ob, capi, quirks):
    q = 0 if quirks == "intent" else (capi.USQ_REF_REFLECT | capi.USQ_UNIT_GGX_PDF)
    sc = mi.load_file(scene_path("us_sphere_box.xml"), paths_per_ray=100, seed=4)
    ui = sc.integrator()
    ui.quirks = q
    ui.simulate_acquisition_parallel(sc)
    ref, _ = ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc), 4, 100)
    check(ui.channel_buf, ref)


def test_path_sharding_adds_up(mi):
    """paths [0,P) == paths [0,a) + [a,P) when every shard is normalised by the total (multi-GPU contract)."""
    sc = mi.load_file(scene_path("us_plate.xml"), seed=6)
    ui = sc.integrator()
    full = ui._acquire(sc, ui.quirks, paths_per_ray=300, seed=6)
    a = ui._acquire(sc, ui.quirks, paths_per_ray=100, path_offset=0, norm_paths=300, seed=6)
    b = ui._acquire(sc, ui.quirks, paths_per_ray=200, path_offset=100, norm_paths=300, seed=6)
    assert rel_l2(a + b, full) <= 1e-5 and np.array_equal((a + b) != 0, full != 0)


def test_roughness_update_changes_the_acquisition(mi, ob):
    """USMain.py:262-265: params['shape.bsdf.roughness'] = v; params.update(); re-run."""
    sc = mi.load_file(scene_path("us_plate.xml"), paths_per_ray=64, seed=0)
    ui = sc.integrator()
    ui.simulate_acquisition_parallel(sc)
    before = ui.channel_buf.copy()
    params = mi.traverse(sc)
    params["shape.bsdf.roughness"] = 0.1
    params.update()
    ui.simulate_acquisition_parallel(sc)
    ref, _ = ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc), 0, 64)
    assert not np.array_equal(before, ui.channel_buf)
    check(ui.channel_buf, ref)


def test_config3_scale_properties(mi):
    """BASELINE config 3 geometry at a large path count: determinism of the non-zero pattern, finite values,
    energy scales with 1/P normalisation (mean over disjoint path ranges agree within Monte-Carlo noise)."""
    sc = mi.load_file(scene_path("us_sphere_box.xml"), seed=0)
    ui = sc.integrator()
    P = 4096
    a = ui._acquire(sc, ui.quirks, paths_per_ray=P, path_offset=0, seed=0)
    b = ui._acquire(sc, ui.quirks, paths_per_ray=P, path_offset=P, seed=0)
    st = mi.default_context().stats()
    assert st["samples"] == 5 * 64 * P and np.isfinite(a).all() and np.isfinite(b).all()
    # the literal estimator multiplies by pdf = 1 / (4 |cos|) (quirks B9 / A7), which is unbounded: single paths put
    # 10^3 times the typical echo into a bin, so plain sums of disjoint path ranges do not agree.  The energy below the
    # 99th percentile of the non-zero bins does (oracle at P = 1024: 2.129 vs 2.116), and so does the set of bins.
    thr = np.quantile(np.abs(a[a != 0]), 0.99)
    ea, eb = float(np.abs(a)[np.abs(a) <= thr].sum()), float(np.abs(b)[np.abs(b) <= thr].sum())
    assert ea > 0 and abs(ea - eb) / ea < 0.03
    assert abs(int((a != 0).sum()) - int((b != 0).sum())) <= 0.01 * (a != 0).sum()
    again = ui._acquire(sc, ui.quirks, paths_per_ray=P, path_offset=0, seed=0)
    assert np.array_equal(again != 0, a != 0) and rel_l2(again, a) <= 1e-5


@pytest.mark.parametrize("seed,n_spheres,n_plates,quirks", [(1, 1, 2, None), (2, 3, 4, 0), (3, 2, 40, None)])
def test_random_phantoms_match_oracle(mi, ob, capi, seed, n_spheres, n_plates, quirks):
    """random phantoms in front of the probe: a few shapes (brute force) and many plates (BVH), the reference's literal
    arithmetic (default quirks) and the intent arithmetic (quirks = 0)"""
    rng = np.random.default_rng(seed)
    T = mi.ScalarTransform4f
    d = {"type": "scene",
         "integrator": {"type": "ultrasound_integrator", "max_depth": 6, "sampling_rate": 40e6, "frequency": 4e6, "sound_speed": 1500,
                        "attenuation": 0.3, "main_beam_angle": 20, "cutoff_angle": 35, "n_elements": 32, "pitch": 2e-4,
                        "time_samples": 4000, "angles": [-10.0, 0.0, 10.0], "paths_per_ray": 60, "seed": seed,
                        **({} if quirks is None else {"quirks": quirks})},
         "sensor": {"type": "ultrasound_sensor", "to_world": T().look_at([0, 0, 0], [0, 0, 0.03], [0, 1, 0])}}
    for i in range(n_spheres):
        d[f"s{i}"] = {"type": "sphere", "center": [float(rng.uniform(-0.01, 0.01)), float(rng.uniform(-0.005, 0.005)), float(rng.uniform(0.02, 0.05))],
                      "radius": float(rng.uniform(0.003, 0.008)),
                      "bsdf": {"type": "ultrasound_bsdf", "impedance": float(rng.uniform(2, 8)), "roughness": float(rng.uniform(0.2, 0.9))}}
    for i in range(n_plates):
        tw = T().translate([float(rng.uniform(-0.012, 0.012)), float(rng.uniform(-0.004, 0.004)), float(rng.uniform(0.015, 0.06))]) @ \
            T().rotate([0, 1, 0], float(rng.uniform(-40, 40))) @ T().rotate([1, 0, 0], float(rng.uniform(150, 210))) @ \
            T().scale([float(rng.uniform(0.002, 0.01)), float(rng.uniform(0.004, 0.01)), 1])
        d[f"p{i}"] = {"type": "rectangle", "to_world": tw,
                      "bsdf": {"type": "ultrasound_bsdf", "impedance": float(rng.uniform(2, 8)), "roughness": float(rng.uniform(0.2, 0.9))}}
    sc = mi.load_dict(d)
    assert (len(sc.flatten()["prims"]) > 32) == (n_plates > 30)
    ui = sc.integrator()
    ui.simulate_acquisition_parallel(sc)
    ref, _ = ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc), seed, 60)
    check(ui.channel_buf, ref)
    assert (ref != 0).sum() > 50


@pytest.mark.parametrize("scene,kw", [("us_sphere_box.xml", {}), ("us_cone_box.xml", {}), ("us_cone_box.xml", dict(tessellate="true"))])
def test_first_bounce_tables_change_nothing(mi, capi, scene, kw):
    """the shared first hit / visibility tables (k_us_first) against every path walking the scene itself: the same
    arithmetic, so the same channel buffer up to the order of the float additions"""
    sc = mi.load_file(scene_path(scene), paths_per_ray=128, seed=8, **kw)
    ui = sc.integrator()
    with_tables = ui._acquire(sc, ui.quirks)
    without = ui._acquire(sc, ui.quirks | capi.USQ_NO_FIRST_TABLES)
    assert np.array_equal(with_tables != 0, without != 0) and (with_tables != 0).sum() > 100
    assert np.allclose(with_tables, without, rtol=2e-5, atol=1e-7 * np.abs(without).max())


def _plate_stack(mi, n_plates, tessellated):
    """steel plates behind each other, growing with depth, reference arithmetic: paths stay alive for several bounces and
    the second bounce still reaches the probe past the smaller plates in front (tilted by 3 degrees and more: at exactly
    normal incidence the reference's GGX frame divides 0 by 0)"""
    T = mi.ScalarTransform4f
    d = {"type": "scene",
         "integrator": {"type": "ultrasound_integrator", "max_depth": 8, "sampling_rate": 40e6, "frequency": 4e6, "sound_speed": 1500,
                        "attenuation": 0.1, "main_beam_angle": 20, "cutoff_angle": 35, "n_elements": 32, "pitch": 2e-4,
                        "time_samples": 4000, "angles": [-5.0, 0.0, 5.0], "paths_per_ray": 300, "seed": 9},
         "sensor": {"type": "ultrasound_sensor", "to_world": T().look_at([0, 0, 0], [0, 0, 0.03], [0, 1, 0])}}
    for i in range(n_plates):
        tw = T().translate([0, 0, 0.015 + 0.006 * i]) @ T().rotate([1, 0, 0], 183 + 3 * i) @ T().scale([0.002 + 0.003 * i, 0.03, 1])
        d[f"p{i}"] = {"type": "rectangle", "to_world": tw, "bsdf": {"type": "ultrasound_bsdf", "impedance": 7.8, "roughness": 0.9}}
    if tessellated:   # > 32 primitives: the BVH kernels (per-wave regions)
        d["c"] = {"type": "cone", "tessellate": True, "segments": 16, "rings": 2,
                  "to_world": T().translate([0, 0, 0.08]) @ T().scale([0.01, 0.01, 0.01]),
                  "bsdf": {"type": "ultrasound_bsdf", "impedance": 7.8, "roughness": 0.9}}
    return mi.load_dict(d)


@pytest.mark.parametrize("case", ["us_plate.xml", "us_sphere_box.xml", "us_cone_box.xml", "us_cone_box.xml:mesh", "stack", "stack:mesh"])
def test_fused_bounces_change_nothing(mi, ob, capi, case):
    """all bounces of a pass in one launch (every region / wave carries its own survivors on) against one launch per
    bounce: the same paths, the same echoes, the same per-depth live counts; float additions in another order"""
    name, _, mesh = case.partition(":")
    if name == "stack":
        sc = _plate_stack(mi, 6, bool(mesh))
    else:
        sc = mi.load_file(scene_path(name), paths_per_ray=256, seed=4, **(dict(tessellate="true") if mesh else {}))
    assert (len(sc.flatten()["prims"]) > 32) == bool(mesh)
    ui = sc.integrator()
    fused = ui._acquire(sc, ui.quirks)
    st_f = mi.default_context().stats()
    per_bounce = ui._acquire(sc, ui.quirks | capi.USQ_NO_FUSED_BOUNCES)
    st_p = mi.default_context().stats()
    assert st_f["live"] == st_p["live"] and st_f["segments"] == st_p["segments"]
    if mesh:   # BVH scenes run k_trace / k_us_shade whatever the switch says: tables at depth 0, two launches per later bounce, a flush
        assert st_f["bounce_launches"] == st_p["bounce_launches"] == 1 + 2 * (ui.max_depth - 1) + 2
    else:
        assert st_f["bounce_launches"] == 1 and st_p["bounce_launches"] == ui.max_depth
    assert np.array_equal(fused != 0, per_bounce != 0) and (fused != 0).sum() > 100
    assert np.allclose(fused, per_bounce, rtol=2e-5, atol=1e-7 * np.abs(per_bounce).max())
    if name == "stack":     # the loop really runs: paths alive at bounces 1, 2 and 3 -- and the oracle agrees
        assert st_f["live"][1] > 1000 and st_f["live"][2] > 100 and st_f["live"][3] > 10
        ref, _ = ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc), 9, 300)
        check(fused.reshape(ref.shape), ref)


def test_exact_normal_incidence_is_nan_as_in_the_reference(mi, ob):
    """under the reference's literal arithmetic the 0-degree plane wave on a plate facing the probe
    exactly (normal = -z) makes _ggx_sample normalise a zero vector (CustomBSDF.py:32-58) -- NaN echoes, the same bins
    in the oracle and on the GPU; the intent arithmetic (quirks = 0) has no such case."""
    T = mi.ScalarTransform4f

    def scene(quirks):
        return mi.load_dict({
            "type": "scene",
            "integrator": {"type": "ultrasound_integrator", "max_depth": 4, "sampling_rate": 40e6, "frequency": 4e6, "sound_speed": 1500,
                           "attenuation": 0.1, "main_beam_angle": 20, "cutoff_angle": 35, "n_elements": 32, "pitch": 2e-4,
                           "time_samples": 4000, "angles": [-5.0, 0.0, 5.0], "paths_per_ray": 50, "seed": 9,
                           **({} if quirks is None else {"quirks": quirks})},
            "sensor": {"type": "ultrasound_sensor", "to_world": T().look_at([0, 0, 0], [0, 0, 0.03], [0, 1, 0])},
            "p": {"type": "rectangle", "to_world": T().translate([0, 0, 0.015]) @ T().rotate([1, 0, 0], 180) @ T().scale([0.03, 0.03, 1]),
                  "bsdf": {"type": "ultrasound_bsdf", "impedance": 7.8, "roughness": 0.9}}})

    sc = scene(None)
    ui = sc.integrator()
    got = ui._acquire(sc, ui.quirks).reshape(3, 32, 4000)
    ref, _ = ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc), 9, 50)
    nan = np.isnan(ref)
    assert nan.sum() > 100 and np.array_equal(np.isnan(got), nan) and not nan[[0, 2]].any()      # the 0-degree angle only
    assert np.array_equal((got != 0) & ~nan, (ref != 0) & ~nan)
    assert rel_l2(np.where(nan, 0, got), np.where(nan, 0, ref)) <= TOL_REL_L2
    sc = scene(0)
    ui = sc.integrator()
    got = ui._acquire(sc, ui.quirks).reshape(3, 32, 4000)
    ref, _ = ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc), 9, 50)
    check(got, ref)


@pytest.mark.parametrize("scene,kw", [("us_testring.xml", {}), ("us_cone_box.xml", dict(tessellate="true"))])
def test_mesh_phantoms_as_streams_and_as_the_fused_bounce(mi, ob, capi, scene, kw):
    """BVH scenes in ultrasound mode.  Product: k_trace (closest hits + the occlusion rays of the previous bounce) and k_us_shade
    (the echo waits in the path state for its occlusion ray; a flush after the last bounce): equal to the oracle, the same with and
    without the first-bounce tables, in several passes, and with the tree in global memory.  Diagnostic build, PBRT_US_FUSED_BVH=1:
    the fused k_us_bounce on the same tree -- same echoes, same segment and per-depth path counts."""
    import os
    ppr = 192
    sc = mi.load_file(scene_path(scene), paths_per_ray=ppr, seed=12, **kw)
    ui = sc.integrator()
    buf = ui._acquire(sc, ui.quirks)
    st = mi.default_context().stats()
    ref, _ = ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc), 12, ppr)
    check(buf, ref)
    assert st["bounce_launches"] == 1 + 2 * (ui.max_depth - 1) + 2          # tables at depth 0, two launches per later bounce, the flush
    no_tab = ui._acquire(sc, ui.quirks | capi.USQ_NO_FIRST_TABLES)
    st_nt = mi.default_context().stats()
    check(no_tab, ref)
    assert st_nt["bounce_launches"] == 2 * ui.max_depth + 2 and st_nt["segments"] == st["segments"] and list(st_nt["live"]) == list(st["live"])
    halves = sum(ui._acquire(sc, ui.quirks, paths_per_ray=ppr // 2, path_offset=o, norm_paths=ppr).astype(np.float64) for o in (0, ppr // 2))
    assert rel_l2(halves.astype(np.float32), ref) <= TOL_REL_L2
    sc_g = mi.load_file(scene_path(scene), paths_per_ray=ppr, seed=12, **kw)
    sc_g.accel = capi.ACCEL_BVH_GLOBAL
    check(sc_g.integrator()._acquire(sc_g, ui.quirks), ref)
    os.environ["PBRT_US_FUSED_BVH"] = "1"
    try:
        with capi.use_library(capi.DIAG_LIB_PATH):
            sc_d = mi.load_file(scene_path(scene), paths_per_ray=ppr, seed=12, **kw)
            fused = sc_d.integrator()._acquire(sc_d, ui.quirks)
            st_f = mi.default_context().stats()
            sc_d._dev = None
    finally:
        os.environ.pop("PBRT_US_FUSED_BVH", None)
    check(fused, ref)
    assert st_f["bounce_launches"] == 1 and st_f["segments"] == st["segments"] and list(st_f["live"]) == list(st["live"])


@pytest.mark.parametrize("case", ["intent", "drjit", "depth1", "depth2_no_tables", "pulse"])
def test_mesh_phantom_variants_of_the_acquisition_loop(mi, ob, capi, case):
    """the stream kernels (k_trace + k_us_shade) under the switches of the acquisition loop, on the ring phantom, against the oracle:
    intent arithmetic (quirks = 0), the Dr.Jit variant (frozen draws, clamped bins, no tof accumulation, signed roulette),
    max_depth 1 (every path ends at its first bounce: with fewer paths per ray than elements there are no tables, so every echo waits
    for its occlusion ray in a record of an ended path and is deposited by the flush), max_depth 2 without tables (pending echoes of
    survivors and of ended paths together), and echoes without the carrier (pulse model)."""
    ppr = {"depth1": 24, "depth2_no_tables": 80}.get(case, 96)
    sc = mi.load_file(scene_path("us_testring.xml"), paths_per_ray=ppr, seed=21)
    ui = sc.integrator()
    q = ui.quirks
    if case == "intent":
        q = 0
    elif case == "drjit":
        q |= capi.USQ_DRJIT_VARIANT
    elif case == "depth1":
        ui.max_depth = 1
    elif case == "depth2_no_tables":
        ui.max_depth = 2
        q |= capi.USQ_NO_FIRST_TABLES
    elif case == "pulse":
        q |= capi.USQ_NO_CARRIER
    buf = ui._acquire(sc, q, pulse=False)
    st = mi.default_context().stats()
    ref, _ = ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc, q), 21, ppr)
    check(buf, ref)
    assert (buf != 0).sum() > 500
    if case == "depth1":
        assert st["live"][1] == 0 and st["bounce_launches"] == 2 + 2     # trace + shade of bounce 0 (the rays come from k_us_init_wf), then the flush


def test_emitter_primary_rays_on_a_mesh_phantom_and_their_refusals(mi, ob, capi):
    """PBRT_US_PRIMARY_EMITTER on a BVH scene (k_us_init_wf writes the emitter's rays, then k_trace + k_us_shade), the quirk
    switches with it, what it changes against the integrator's own rays, and the error codes: an emitter whose array is not the
    integrator's, a scene without one."""
    T = mi.ScalarTransform4f
    sc = mi.load_file(scene_path("us_testring.xml"), paths_per_ray=48, seed=6)
    ui = sc.integrator()
    own = ui._acquire(sc, ui.quirks)
    em = mi.CustomEmitter(mi.Properties("ultrasound_emitter", dict(number_of_elements=ui.n_elements, pitch=ui.pitch, element_width=1e-4,
                                                                   element_height=4e-4, number_of_rays_per_element=48, speed_of_sound=ui.sound_speed,
                                                                   steering_angle_min=-12.0, steering_angle_max=12.0)))
    sc._emitters.append(em)
    ui.primary_rays = "emitter"
    p = ui.us_params(sc)
    assert p.primary == capi.US_PRIMARY_EMITTER and p.emitter.number_of_elements == ui.n_elements
    got = ui._acquire(sc, ui.quirks)
    ref, tx = ob.OracleScene.from_scene(sc).us_acquire(p, 6, 48)
    check(got, ref)
    assert np.array_equal(ui.transmission_delays_buf, tx)                  # the table of the nominal angles, as before
    assert np.abs(got).max() > 0 and not np.array_equal(got != 0, own != 0)   # other rays, other echoes
    # the emitter's weight max(0, d.n) / N_total_rays is the path's amplitude: the echoes are ~ 1 / (64 * 48) of the integrator's own
    assert np.abs(got).max() < 0.01 * np.abs(own).max()
    q = ui.quirks | capi.USQ_DRJIT_VARIANT
    check(ui._acquire(sc, q), ob.OracleScene.from_scene(sc).us_acquire(ui.us_params(sc, q), 6, 48)[0])
    em.number_of_elements = ui.n_elements // 2
    with pytest.raises(ValueError, match="elements"):
        ui.us_params(sc)
    p.emitter.number_of_elements = 7                                          # past the Python check: the library refuses it
    dev = sc.device()
    buf = np.empty((ui.n_angles, ui.n_elements, ui.time_samples), np.float32)
    rc = dev.ctx.lib.pbrt_us_acquire(dev.handle, p, 6, 4, 0, 4, buf.ctypes.data, None)
    assert rc == -1 and b"number_of_elements" in dev.ctx.lib.pbrt_last_error(dev.ctx.handle)
    sc._emitters.clear()
    with pytest.raises(ValueError, match="ultrasound_emitter"):
        ui.us_params(sc)


@pytest.mark.parametrize("scene,ppr", [("us_sphere_box.xml", 128), ("us_sphere_box.xml", 24), ("us_plate.xml", 200)])
def test_specialised_and_generic_bounce_kernels_agree(mi, capi, monkeypatch, scene, ppr):
    """k_us_bounce exists with the library's default switches compiled in (PBRT_USQ_REFERENCE, with / without the carrier; with /
    without the first-bounce tables) and as a generic instance that reads pbrt_us_params.quirks at run time (any other set).
    PBRT_US_GENERIC_KERNEL=1 sends the default set through the generic instance: same segments, same echoes."""
    sc = mi.load_file(scene_path(scene), paths_per_ray=ppr, seed=21)
    ui = sc.integrator()
    ctx = mi.default_context()
    for q in (ui.quirks, ui.quirks | capi.USQ_NO_CARRIER, ui.quirks | capi.USQ_NO_FIRST_TABLES):
        monkeypatch.delenv("PBRT_US_GENERIC_KERNEL", raising=False)
        fast = ui._acquire(sc, q, pulse=False)
        st_fast = ctx.stats()
        monkeypatch.setenv("PBRT_US_GENERIC_KERNEL", "1")
        gen = ui._acquire(sc, q, pulse=False)
        st_gen = ctx.stats()
        monkeypatch.delenv("PBRT_US_GENERIC_KERNEL")
        assert st_fast["segments"] == st_gen["segments"] > 0 and st_fast["live"] == st_gen["live"]
        assert np.array_equal(fast != 0, gen != 0) and np.allclose(fast, gen, rtol=2e-5, atol=1e-7 * np.abs(gen).max())


def test_emitter_rays_with_and_without_the_region_permutation(mi, ob, monkeypatch):
    """PBRT_US_PRIMARY_EMITTER at a size where a pass has many regions (160 regions of 8192 paths): workgroup b walks region
    (b * m) mod regions (a permutation; kernels_us.h) -- the same echoes as with workgroup b on region b (f32 atomics in another
    order), the same counters, and the oracle's channel buffer at the acquisition's tolerance"""
    sc = mi.load_file(scene_path("us_sphere_box.xml"), primary_rays="emitter", paths_per_ray=4096, seed=6)
    ui = sc.integrator()
    ctx = sc.device().ctx
    a = ui._acquire(sc, ui.quirks)
    st_a = ctx.stats()
    monkeypatch.setenv("PBRT_US_EMIT_PERMUTE", "0")
    b = ui._acquire(sc, ui.quirks)
    st_b = ctx.stats()
    monkeypatch.delenv("PBRT_US_EMIT_PERMUTE")
    monkeypatch.setenv("PBRT_US_EMIT_PERMUTE", "7")                      # another stride
    c = ui._acquire(sc, ui.quirks)
    monkeypatch.delenv("PBRT_US_EMIT_PERMUTE")
    monkeypatch.setenv("PBRT_US_EMIT_FUSED", "0")                        # the rays through k_us_emit_init and the path state
    e = ui._acquire(sc, ui.quirks)
    st_e = ctx.stats()
    monkeypatch.delenv("PBRT_US_EMIT_FUSED")
    assert (st_a["segments"], st_a["shadow_rays"], st_a["live"]) == (st_b["segments"], st_b["shadow_rays"], st_b["live"])
    assert (st_a["segments"], st_a["shadow_rays"], st_a["live"]) == (st_e["segments"], st_e["shadow_rays"], st_e["live"])
    for x in (b, c, e):
        assert np.array_equal(a != 0, x != 0) and np.allclose(a, x, rtol=0, atol=2e-5 * np.abs(a).max())
    osc = ob.OracleScene.from_scene(sc)
    ref, _ = osc.us_acquire(ui.us_params(sc), 6, 4096)
    ref = np.asarray(ref).reshape(a.shape)
    assert np.linalg.norm(a - ref) <= 1e-3 * np.linalg.norm(ref) and np.array_equal(a != 0, ref != 0)
