"""K8 on the DEVICE: closed forms that need no oracle at all -- white furnace, direct illumination of a
plane by a small parallel light, mirror / Snell / Fresnel / total-internal-reflection directions, cosine-hemisphere moments,
the arrival bin of a single-plate echo -- run through libpbrt_hip.so (ctypes, the plugin API).  The same scenes and tolerances
as tests/test_oracle_transport.py:29-145, where they pin the CPU restatement; here nothing under oracle/ is imported, so these
checks hold whatever the oracle does.  (The oracle is pinned by no reference output: DESIGN.md section 2.)"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rect(T, bsdf=None, emitter=None):
    d = {"type": "rectangle", "to_world": T}
    if bsdf is not None:
        d["bsdf"] = bsdf
    if emitter is not None:
        d["emitter"] = emitter
    return d


def _camera(mi, origin, target, up, res, spp, fov=40.0, filt="box"):
    return {"type": "perspective", "fov": fov, "near_clip": 1e-3, "far_clip": 100.0,
            "to_world": mi.ScalarTransform4f().look_at(origin, target, up),
            "sampler": {"type": "independent", "sample_count": spp},
            "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": filt}}}


@pytest.mark.parametrize("accel", ["brute", "bvh", "bvh_global"])
def test_white_furnace_on_the_device(mi, capi, accel):
    """Closed diffuse box with albedo rho whose six walls all emit Le: a path of depth budget D carries
    L = Le (1 + rho + ... + rho^(D-1)); emitter hits and emitter sampling are MIS-weighted against each other and together give Le
    per vertex, so the film is that constant up to Monte-Carlo noise in the MIS split.  Brute-force kernel (k_bounce chain), the
    LDS-resident BVH and the BVH in global memory (k_trace_primary + k_trace + k_shade)."""
    T = mi.ScalarTransform4f
    rho, Le, D = 0.5, 0.75, 5
    bs = {"type": "diffuse", "reflectance": {"type": "rgb", "value": [rho] * 3}}
    em = lambda: {"type": "area", "radiance": {"type": "rgb", "value": [Le] * 3}}
    walls = {
        "zp": T().translate([0, 0, 1]).rotate([0, 1, 0], 180), "zn": T().translate([0, 0, -1]),
        "xp": T().translate([1, 0, 0]).rotate([0, 1, 0], -90), "xn": T().translate([-1, 0, 0]).rotate([0, 1, 0], 90),
        "yp": T().translate([0, 1, 0]).rotate([1, 0, 0], 90), "yn": T().translate([0, -1, 0]).rotate([1, 0, 0], -90)}
    d = {"type": "scene", "integrator": {"type": "path", "max_depth": D, "rr_depth": 100},
         "sensor": _camera(mi, [0, 0, 0], [0.3, 0.2, 1], [0, 1, 0], 16, 64)}
    for k, t in walls.items():
        d[k] = _rect(t, bs, em())
    sc = mi.load_dict(d)
    sc.accel = {"brute": capi.ACCEL_BRUTE, "bvh": capi.ACCEL_BVH, "bvh_global": capi.ACCEL_BVH_GLOBAL}[accel]
    for p in sc.flatten()["prims"]:   # all normals face the centre
        c = p["g"][0:3] + 0.5 * p["g"][3:6] + 0.5 * p["g"][6:9]
        assert np.dot(p["g"][9:12], -c) > 0.99
    img = mi.render(sc, seed=0, spp=64)
    want = Le * sum(rho ** k for k in range(D))
    assert img.shape == (16, 16, 3) and np.isfinite(img).all()
    assert np.allclose(img.mean(axis=(0, 1)), want, rtol=2e-2)
    assert np.abs(img - want).max() < 0.25 * want          # every pixel, 64 spp
    # without Russian roulette and with a budget of one bounce the film is the emission itself, exactly
    d["integrator"] = {"type": "path", "max_depth": 1}
    sc1 = mi.load_dict(d)
    sc1.accel = sc.accel
    assert np.array_equal(mi.render(sc1, seed=3, spp=4), np.full((16, 16, 3), Le, np.float32))


@pytest.mark.parametrize("accel", ["brute", "bvh"])
def test_direct_illumination_closed_form_on_the_device(mi, capi, accel):
    """Diffuse floor lit by a small square light straight above: E = Le A / h^2 at the point below it (the exact form factor of a
    parallel square differs from the small-source value by < 0.1 %), L = rho / pi * E."""
    T = mi.ScalarTransform4f
    rho, Le, h, half = 0.8, 50.0, 2.0, 0.05
    d = {"type": "scene", "integrator": {"type": "direct"},
         "sensor": _camera(mi, [2.0, 3.0, 0.0], [0, 0, 0], [0, 0, 1], 9, 256, fov=0.5),
         "floor": _rect(T().rotate([1, 0, 0], -90).scale(10), {"type": "diffuse", "reflectance": {"type": "rgb", "value": [rho] * 3}}),
         "lamp": _rect(T().translate([0, h, 0]).rotate([1, 0, 0], 90).scale(half),
                       {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0, 0, 0]}},
                       {"type": "area", "radiance": {"type": "rgb", "value": [Le] * 3}})}
    sc = mi.load_dict(d)
    sc.accel = capi.ACCEL_BRUTE if accel == "brute" else capi.ACCEL_BVH
    P = sc.flatten()["prims"]
    assert np.allclose(P[0]["g"][9:12], [0, 1, 0], atol=1e-6) and np.allclose(P[1]["g"][9:12], [0, -1, 0], atol=1e-6)
    img = mi.render(sc, seed=0, spp=256)
    want = rho / math.pi * Le * (2 * half) ** 2 / (h * h)
    assert img[4, 4].mean() == pytest.approx(want, rel=2e-2)


def test_mirror_and_glass_directions_through_pbrt_bsdf_sample(mi):
    ctx = mi.BSDFContext()
    mirror = mi.ConductorBSDF(mi.Properties("conductor"))
    wi = np.array([[0.3, -0.2, math.sqrt(1 - 0.13)]], np.float32)
    bs, w = mirror.sample(ctx, mi.SurfaceInteraction3f(wi), 0.5, [[0.5, 0.5]])
    assert np.allclose(bs.wo[0], [-0.3, 0.2, wi[0, 2]]) and bs.pdf[0] == 1 and np.allclose(w[0], 1)
    # Snell at BK7 / air: sin(theta_t) = sin(theta_i) / eta
    eta = 1.5046 / 1.000277
    glass = mi.DielectricBSDF(mi.Properties("dielectric"))
    assert glass.eta == pytest.approx(eta, rel=1e-12)
    th = math.radians(40.0)
    wi = np.array([[math.sin(th), 0, math.cos(th)]], np.float32)
    bs, w = glass.sample(ctx, mi.SurfaceInteraction3f(wi), 0.999, [[0.5, 0.5]])       # s1 > F: refraction
    assert bs.sampled_component[0] == 1 and bs.wo[0, 2] < 0
    assert math.hypot(bs.wo[0, 0], bs.wo[0, 1]) == pytest.approx(math.sin(th) / eta, rel=1e-5)
    assert np.linalg.norm(bs.wo[0]) == pytest.approx(1.0, rel=1e-6)
    assert w[0, 0] == pytest.approx(1 / eta ** 2, rel=1e-5)                            # radiance scaling eta_ti^2
    # Fresnel reflectance at normal incidence ((eta - 1) / (eta + 1))^2
    bs, w = glass.sample(ctx, mi.SurfaceInteraction3f(np.array([[0, 0, 1.0]], np.float32)), 0.0, [[0.5, 0.5]])
    assert bs.sampled_component[0] == 0 and bs.pdf[0] == pytest.approx(((eta - 1) / (eta + 1)) ** 2, rel=1e-5)
    # total internal reflection from inside at 60 degrees
    th = math.radians(60.0)
    bs, w = glass.sample(ctx, mi.SurfaceInteraction3f(np.array([[math.sin(th), 0, -math.cos(th)]], np.float32)), 0.999, [[0.5, 0.5]])
    assert bs.sampled_component[0] == 0 and bs.pdf[0] == pytest.approx(1.0)


def test_cosine_hemisphere_moments_on_the_device(mi):
    rng = np.random.default_rng(1)
    n = 200000
    b = mi.DiffuseBSDF(mi.Properties("diffuse", dict(reflectance=[0.5, 0.6, 0.7])))
    wi = np.tile(np.array([[0, 0, 1.0]], np.float32), (n, 1))
    si = mi.SurfaceInteraction3f(wi)
    bs, w = b.sample(mi.BSDFContext(), si, 0.5, rng.random((n, 2), dtype=np.float32))
    wo, pdf = bs.wo, bs.pdf
    assert np.allclose(np.linalg.norm(wo, axis=1), 1, atol=1e-5) and np.all(wo[:, 2] >= 0)
    assert wo[:, 2].mean() == pytest.approx(2 / 3, abs=3e-3)        # E[cos] under a cosine density
    assert np.allclose(pdf, wo[:, 2] / math.pi, atol=1e-6) and np.allclose(w, [[0.5, 0.6, 0.7]])
    f, p2 = b.eval_pdf(None, si, wo)
    assert np.allclose(p2, pdf, atol=1e-7) and np.allclose(f[:, 1], 0.6 * wo[:, 2] / math.pi, atol=1e-6)


@pytest.mark.parametrize("tables", [True, False])
def test_single_plate_echo_arrival_bins_on_the_device(mi, capi, tables):
    """Plate perpendicular to the beam at depth z: the first echo of element e received at element r arrives at
    t = z / c + sqrt(z^2 + (x_r - x_e)^2) / c  ->  bin round(t fs)  (CustomIntegrator.py:316,329,351-352); the on-axis echo at
    round(2 z / c fs).  With and without the first-bounce tables (k_us_first + table-driven depth 0 / every path walks)."""
    T = mi.ScalarTransform4f
    z, c, fs, N, pitch = 0.03, 1540.0, 50e6, 16, 3e-4
    d = {"type": "scene",
         "integrator": {"type": "ultrasound_integrator", "max_depth": 1, "sampling_rate": fs, "frequency": 5e6, "sound_speed": c,
                        "attenuation": 0.0, "main_beam_angle": 80, "cutoff_angle": 85, "n_elements": N, "pitch": pitch,
                        "time_samples": 4000, "angles": np.array([0.0], np.float32)},
         "sensor": {"type": "ultrasound_sensor", "to_world": T().look_at([0, 0, 0], [0, 0, 0.03], [0, 1, 0])},
         "plate": {"type": "rectangle", "to_world": T().translate([0, 0, z]).rotate([0, 1, 0], 180).scale(0.5),
                   "bsdf": {"type": "ultrasound_bsdf", "impedance": 7.8, "roughness": 0.7}}}
    sc = mi.load_dict(d)
    ui = sc.integrator()
    q = ui.quirks | (0 if tables else capi.USQ_NO_FIRST_TABLES)
    buf = ui._acquire(sc, q, paths_per_ray=400, seed=5)
    assert np.all(ui.transmission_delays_buf == 0) and buf.shape == (1, N, 4000)
    ex = ui.elem_x.numpy()
    allowed = np.zeros((N, 4000), bool)
    for r in range(N):
        for e in range(N):
            t = z / c + math.sqrt(z * z + (ex[r] - ex[e]) ** 2) / c
            allowed[r, int(np.rint(np.float32(t) * np.float32(fs)))] = True
            allowed[r, min(3999, int(np.rint(t * fs)) + 1)] = True
            allowed[r, int(np.rint(t * fs)) - 1] = True
    nz = buf[0] != 0
    assert nz.sum() > 0 and not np.any(nz & ~allowed)
    assert np.argwhere(nz)[:, 1].min() == int(np.rint(2 * z / c * fs))      # on-axis echo: 2 z / c


def test_emitter_rays_without_jitter_are_the_integrators_own_rays(mi, capi):
    """PBRT_US_PRIMARY_EMITTER against the acquisition it generalises, no oracle involved: an emitter with point elements
    (element_width = element_height = 0) and ONE steering angle of 0 degrees draws exactly the integrator's own primary ray of the
    0-degree transmission (origin (x_e, 0, 0), direction +z, emission time 0 = tx_delay), so every path is the same path -- same
    bounces, same draws -- and every echo is the integrator's echo times the ray's weight max(0, d.n) / (N rays_per_element) =
    1 / (16 * 3).  One side reads the first-bounce tables, the other traces every path from its own origin."""
    T = mi.ScalarTransform4f
    N, ppr = 16, 96
    d = {"type": "scene",
         "integrator": {"type": "ultrasound_integrator", "max_depth": 4, "sampling_rate": 50e6, "frequency": 3e6, "sound_speed": 1480.0,
                        "attenuation": 0.1, "main_beam_angle": 24, "cutoff_angle": 30, "n_elements": N, "pitch": 3e-4,
                        "time_samples": 6000, "angles": np.array([0.0], np.float32), "paths_per_ray": ppr, "seed": 4},
         "sensor": {"type": "ultrasound_sensor", "to_world": T().look_at([0, 0, 0], [0, 0, 0.03], [0, 1, 0])},
         "emitter": {"type": "ultrasound_emitter", "number_of_elements": N, "pitch": 3e-4, "element_width": 0.0, "element_height": 0.0,
                     "number_of_rays_per_element": 3, "speed_of_sound": 1480.0, "steering_angle_min": 0.0, "steering_angle_max": 0.0},
         "plate": {"type": "rectangle", "to_world": T().translate([0.002, 0, 0.02]) @ T().rotate([0, 1, 0], 180 + 9) @ T().scale([0.004, 0.01, 1]),
                   "bsdf": {"type": "ultrasound_bsdf", "impedance": 7.8, "roughness": 0.7}},
         # (tilted: a surface met at EXACTLY normal incidence gives NaN echoes under the reference's literal arithmetic, DESIGN D12)
         "wall": {"type": "rectangle", "to_world": T().translate([0, 0, 0.05]) @ T().rotate([0, 1, 0], 180 - 4) @ T().scale([0.05, 0.05, 1]),
                  "bsdf": {"type": "ultrasound_bsdf", "impedance": 7.8, "roughness": 0.5}}}
    sc = mi.load_dict(d)
    ui = sc.integrator()
    own = ui._acquire(sc, ui.quirks)
    st_own = mi.default_context().stats()
    ui.primary_rays = "emitter"
    assert ui.us_params(sc).primary == capi.US_PRIMARY_EMITTER
    em = ui._acquire(sc, ui.quirks)
    st_em = mi.default_context().stats()
    w = 1.0 / (N * 3)
    # "the same path" up to the last bit of the element positions: the emitter places its elements with linspace
    # (CustomEmmitter.py:34-36), the integrator with pitch * (i - (N - 1) / 2) (CustomIntegrator.py:28-30) -- so a roulette or
    # time-bin decision may fall the other way for a path in a million: the counts agree to 1e-3, the buffers to 1e-3 in L2
    assert st_own["segments"] >= N * ppr and abs(st_em["segments"] - st_own["segments"]) <= 1e-3 * st_own["segments"]
    assert all(abs(a - b) <= 1e-3 * max(b, 1) + 2 for a, b in zip(st_em["live"], st_own["live"]))
    assert np.isfinite(own).all() and np.abs(own).max() > 0 and np.mean((em != 0) == (own != 0)) > 0.9999
    ref = own.astype(np.float64) * w
    assert np.linalg.norm(em - ref) <= 1e-3 * np.linalg.norm(ref)
