"""Randomised geometry: spheres, rectangles (parallelograms), analytic cones and triangle meshes in random poses, small
scenes (brute force) and large ones (BVH), random rays -- the closest hit, the occlusion test and a short render equal the oracle's
bit for bit.  Seeds are fixed: the cases are reproducible."""
import numpy as np
import pytest

from conftest import oracle_render, scene_path

pytestmark = pytest.mark.gpu


def _random_scene(mi, tmp_path, seed, n_spheres, n_rects, n_tris, n_cones=0):
    rng = np.random.default_rng(seed)
    T = mi.ScalarTransform4f
    d = {"type": "scene", "integrator": {"type": "path", "max_depth": 4},
         "sensor": {"type": "perspective", "fov": 50, "to_world": T().look_at([0, 0, 6], [0, 0, 0], [0, 1, 0]),
                    "film": {"type": "hdrfilm", "width": 24, "height": 20, "rfilter": {"type": "tent"}},
                    "sampler": {"type": "independent", "sample_count": 3}},
         "light": {"type": "rectangle", "to_world": T().translate([0, 2.9, 0]) @ T().rotate([1, 0, 0], 90) @ T().scale([1.5, 1.5, 1]),
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [3, 3, 3]}}, "bsdf": {"type": "diffuse"}},
         "floor": {"type": "rectangle", "to_world": T().translate([0, -3, 0]) @ T().rotate([1, 0, 0], -90) @ T().scale([4, 4, 1]),
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.6, 0.6, 0.6]}}}}
    mats = [{"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.7, 0.4, 0.3]}}, {"type": "conductor"}, {"type": "dielectric"}]
    for i in range(n_spheres):
        d[f"s{i}"] = {"type": "sphere", "center": list(rng.uniform(-2, 2, 3)), "radius": float(rng.uniform(0.2, 0.7)), "bsdf": mats[i % 3]}
    for i in range(n_rects):
        tw = T().translate(list(rng.uniform(-2, 2, 3))) @ T().rotate(list(rng.normal(size=3)), float(rng.uniform(0, 360))) @ \
            T().scale([float(rng.uniform(0.3, 1.2)), float(rng.uniform(0.3, 1.2)), 1])
        d[f"r{i}"] = {"type": "rectangle", "to_world": tw, "bsdf": mats[0]}
    for i in range(n_cones):   # any affine pose: non-uniform scale, every other one mirrored
        sx = float(rng.uniform(0.3, 0.9)) * (-1 if i % 2 else 1)
        tw = T().translate(list(rng.uniform(-2, 2, 3))) @ T().rotate(list(rng.normal(size=3)), float(rng.uniform(0, 360))) @ \
            T().scale([sx, float(rng.uniform(0.3, 0.9)), float(rng.uniform(0.4, 1.5))])
        d[f"c{i}"] = {"type": "cone", "to_world": tw, "bsdf": mats[i % 3]}
    if n_tris:
        v = rng.uniform(-2.5, 2.5, (n_tris, 1, 3)) + rng.normal(scale=0.25, size=(n_tris, 3, 3))
        path = tmp_path / f"soup{seed}.obj"
        with open(path, "w") as f:
            for p in v.reshape(-1, 3):
                f.write(f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}\n")
            for i in range(n_tris):
                f.write(f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}\n")
        d["soup"] = {"type": "obj", "filename": str(path), "bsdf": mats[0]}
    return mi.load_dict(d)


@pytest.mark.parametrize("seed,ns,nr,nt,nc", [(1, 3, 4, 0, 0), (2, 5, 10, 12, 0), (3, 0, 2, 400, 0), (4, 6, 6, 1500, 0),
                                              (5, 2, 4, 0, 4), (6, 3, 5, 300, 5)])     # cones: brute force / in BVH leaves
def test_random_scene_leaf_ops_and_render(mi, ob, capi, tmp_path, seed, ns, nr, nt, nc):
    sc = _random_scene(mi, tmp_path, seed, ns, nr, nt, nc)
    n_prims = len(sc.flatten()["prims"])
    assert (n_prims <= 32) == (nt <= 12)                       # both accelerators are exercised across the cases
    rng = np.random.default_rng(100 + seed)
    n = 30000
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:50] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 50)] * rng.choice([-1, 1], (50, 1)).astype(np.float32)   # axis-aligned rays
    tmax = np.where(np.arange(n) % 4 == 0, 2.0, np.inf).astype(np.float32)
    osc = ob.OracleScene.from_scene(sc)
    got = sc.ray_intersect(o, d, tmax)
    t, prim, u, v = osc.ray_intersect(o, d, tmax)
    assert np.array_equal(got["prim"], prim) and np.array_equal(got["t"], t)
    assert np.array_equal(got["u"], u) and np.array_equal(got["v"], v)
    assert np.array_equal(sc.ray_test(o, d, tmax), osc.ray_test(o, d, tmax))
    assert 0.2 < got["valid"].mean() < 1.0
    integ, sens = sc.integrator(), sc.sensors()[0]
    img = mi.render(sc, seed=seed)
    ref = osc.render(sens.camera(), integ._film_desc(sc, sens, seed, 3), n_threads=8)
    assert np.array_equal(img, ref) and img.mean() > 0


def test_mesh_larger_than_lds_takes_the_global_bvh_kernels(mi, ob, capi, tmp_path):
    """a 20 000-triangle sphere mesh (2.5 MB of nodes + primitives: no LDS image): k_bounce / k_us_bounce traverse the
    BVH through the vector caches (ACCEL_K_BVH_GLOBAL) -- same contract, bit-exact against the oracle"""
    nu, nv = 100, 100
    th = np.linspace(0, np.pi, nv + 1)[:, None]
    ph = np.linspace(0, 2 * np.pi, nu, endpoint=False)[None, :]
    P = np.stack([np.sin(th) * np.cos(ph), np.cos(th) * np.ones_like(ph), np.sin(th) * np.sin(ph)], axis=-1)     # [nv+1, nu, 3]
    idx = lambda i, j: i * nu + (j % nu) + 1
    path = tmp_path / "ball.obj"
    with open(path, "w") as f:
        for p in P.reshape(-1, 3):
            f.write(f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}\n")
        for i in range(nv):
            for j in range(nu):
                f.write(f"f {idx(i, j)} {idx(i + 1, j + 1)} {idx(i + 1, j)}\n")
                f.write(f"f {idx(i, j)} {idx(i, j + 1)} {idx(i + 1, j + 1)}\n")
    T = mi.ScalarTransform4f
    sc = mi.load_dict({
        "type": "scene", "integrator": {"type": "path", "max_depth": 4},
        "sensor": {"type": "perspective", "fov": 45, "to_world": T().look_at([0, 1, 5], [0, 0, 0], [0, 1, 0]),
                   "film": {"type": "hdrfilm", "width": 40, "height": 32, "rfilter": {"type": "tent"}},
                   "sampler": {"type": "independent", "sample_count": 4}},
        "ball": {"type": "obj", "filename": str(path), "merge_quads": False,
                 "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.7, 0.5, 0.4]}}},
        "floor": {"type": "rectangle", "to_world": T().translate([0, -1.2, 0]) @ T().rotate([1, 0, 0], -90) @ T().scale([5, 5, 1]),
                  "bsdf": {"type": "diffuse"}},
        "light": {"type": "rectangle", "to_world": T().translate([0, 4, 0]) @ T().rotate([1, 0, 0], 90) @ T().scale([1, 1, 1]),
                  "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [10, 10, 10]}}, "bsdf": {"type": "diffuse"}}})
    assert len(sc.flatten()["prims"]) == 20002
    integ, sens = sc.integrator(), sc.sensors()[0]
    img = mi.render(sc, seed=9)
    ref = ob.OracleScene.from_scene(sc).render(sens.camera(), integ._film_desc(sc, sens, 9, 4), n_threads=8)
    assert np.array_equal(img, ref) and img.mean() > 0.01
    big = integ.render(sc, seed=9, spp=4, pass_paths=3000)            # several passes, repack between bounces
    assert np.array_equal(big, img)


@pytest.mark.parametrize("accel", ["auto", "bvh", "bvh_global"])
def test_cbox_with_the_reference_boxes(mi, ob, capi, accel):
    """SURVEY f-4: scenes/meshes/cbox_largebox.obj + cbox_smallbox.obj in the Cornell room (tests/scenes/cbox_boxes.xml),
    64 x 64 x 4 spp, through the brute-force kernel (22 primitives), the LDS-resident BVH and the BVH read through the
    vector caches (PBRT_ACCEL_BVH_GLOBAL: the kernel variant of meshes larger than LDS) -- each bit-exact against the
    oracle with the same accelerator semantics (DESIGN D9)"""
    sc = mi.load_file(scene_path("cbox_boxes.xml"), res=64, spp=4)
    sc.accel = {"auto": capi.ACCEL_AUTO, "bvh": capi.ACCEL_BVH, "bvh_global": capi.ACCEL_BVH_GLOBAL}[accel]
    img = mi.render(sc, seed=2)
    ref, _ = oracle_render(ob, sc, 2, 4)
    assert np.array_equal(img, ref) and img.mean() > 0.05
    if accel != "auto":   # the two accelerators see the same surfaces: films agree up to tie-breaking on shared edges
        sc2 = mi.load_file(scene_path("cbox_boxes.xml"), res=64, spp=4)
        assert np.mean(np.abs(mi.render(sc2, seed=2) - img)) < 1e-3


@pytest.mark.parametrize("bulb", [[0.3, 0.2, 0.5], [3.0, 0.5, 0.0], [0.0, 4.0, 0.0]])
def test_occluder_pruning_changes_nothing_on_the_device(mi, ob, capi, bulb):
    """DESIGN D11 on the HIP side: same film with the occluder list and with shadow segments that walk every primitive,
    for point lights inside and outside the room; both equal the oracle"""
    from test_oracle_transport import _hull_scene
    sc = _hull_scene(mi, bulb)
    integ = sc.integrator()
    a = integ.render(sc, seed=3, spp=4)
    b = integ.render(sc, seed=3, spp=4, flags=capi.FILM_NO_OCCLUDER_PRUNING)
    ref, _ = oracle_render(ob, sc, 3, 4)
    assert np.array_equal(a, b) and np.array_equal(a, ref)
    cb = mi.load_file(scene_path("cbox.xml"), res=64, spp=8)
    assert np.array_equal(cb.integrator().render(cb, seed=1, spp=8),
                          cb.integrator().render(cb, seed=1, spp=8, flags=capi.FILM_NO_OCCLUDER_PRUNING))


@pytest.mark.parametrize("n_lat,n_lon,accel", [(3, 4, "auto"), (8, 12, "auto"), (8, 12, "bvh_global"), (20, 24, "auto"), (26, 26, "auto")])
def test_meshes_with_vertex_normals(mi, ob, capi, tmp_path, n_lat, n_lon, accel):
    """interpolated shading normals (Mitsuba meshes with `vn`): a 16-triangle ball (brute force: a small scene with vertex
    normals takes the _BIG kernel variant) and a 160-triangle one (LDS BVH, and the BVH read through the vector caches),
    radiance and ultrasound mode, bit-exact / in tolerance against the oracle.  912 triangles: an LDS image (~94 KB) that
    leaves room for k_bounce_pool's hit stacks; 1 300 triangles: one (~133 KB) that does not, so bounces >= 1 stay with k_bounce"""
    from mesh_util import write_uv_sphere_obj
    from test_oracle_transport import _lit_ball
    nt = write_uv_sphere_obj(str(tmp_path / "ball.obj"), n_lat=n_lat, n_lon=n_lon, normals=True)
    dif = {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.7, 0.6]}}
    sc = _lit_ball(mi, {"type": "obj", "filename": str(tmp_path / "ball.obj"), "bsdf": dif})
    sc.integrator().max_depth = 3
    if accel != "auto":
        sc.accel = capi.ACCEL_BVH_GLOBAL
    assert len(sc.flatten()["prims"]) == nt and sc.flatten()["vertex_normals"] is not None
    img = mi.render(sc, seed=4)
    ref, _ = oracle_render(ob, sc, 4, 4)
    assert np.array_equal(img, ref) and img.mean() > 0.01
    flat = _lit_ball(mi, {"type": "obj", "filename": str(tmp_path / "ball.obj"), "face_normals": True, "bsdf": dif})
    assert not np.array_equal(mi.render(flat, seed=4), img)
    # ultrasound mode: the shading normal is si.sh_frame.n of CustomIntegrator.py:340,345 and CustomBSDF.py:91
    T = mi.ScalarTransform4f
    us = mi.load_dict({"type": "scene",
                       "integrator": {"type": "ultrasound_integrator", "max_depth": 4, "sampling_rate": 40e6, "frequency": 4e6, "sound_speed": 1500,
                                      "attenuation": 0.3, "main_beam_angle": 20, "cutoff_angle": 35, "n_elements": 16, "pitch": 2e-4,
                                      "time_samples": 3000, "angles": [-5.0, 5.0], "paths_per_ray": 80, "seed": 2},
                       "sensor": {"type": "ultrasound_sensor", "to_world": T().look_at([0, 0, 0], [0, 0, 0.03], [0, 1, 0])},
                       "ball": {"type": "obj", "filename": str(tmp_path / "ball.obj"),
                                "to_world": T().translate([0.001, 0.0005, 0.02]).scale(0.006),
                                "bsdf": {"type": "ultrasound_bsdf", "impedance": 5.0, "roughness": 0.6}}})
    if accel != "auto":
        us.accel = capi.ACCEL_BVH_GLOBAL
    ui = us.integrator()
    ui.simulate_acquisition_parallel(us)
    want, _ = ob.OracleScene.from_scene(us).us_acquire(ui.us_params(us), 2, 80)
    rel = np.linalg.norm(ui.channel_buf.astype(np.float64) - want) / np.linalg.norm(want.astype(np.float64))
    assert rel <= 1e-3 and np.array_equal(ui.channel_buf != 0, want != 0) and np.abs(want).max() > 0


def test_the_references_bunny_on_the_device(mi, ob, capi):
    """scenes/meshes/bunny.ply of the reference (binary little-endian PLY, 69 451 triangles; committed as a
    data asset) through the BVH that does not fit LDS -- k_trace / k_shade with the tree read through the vector caches -- bit for
    bit the oracle's film (its own median-split tree: the result does not depend on the tree), and the same film at another pass
    size."""
    sc = mi.load_file(scene_path("bunny.xml"), res=64, spp=4)
    assert len(sc.flatten()["prims"]) == 69451 + 2
    integ = sc.integrator()
    img = integ.render(sc, seed=11, spp=4)
    st = mi.default_context().stats()
    ref, _ = oracle_render(ob, sc, 11, 4)
    assert np.array_equal(img, ref) and img.mean() > 0.05
    assert st["plan_source"] == capi.PLAN_STREAMS and st["live"][0] == 64 * 64 * 4 and 0 < st["live"][1] < st["live"][0]
    assert np.array_equal(integ.render(sc, seed=11, spp=4, pass_paths=5000), img)


def test_the_references_suzanne_with_its_vertex_normals(mi, ob, capi):
    """scenes/meshes/suzanne.ply of the reference (62 976 triangles, PLY normals + texture coordinates): the interpolated shading
    normals of Mitsuba's Mesh on a mesh larger than LDS, bit for bit the oracle's film"""
    sc = mi.load_file(scene_path("suzanne.xml"), res=64, spp=4)
    f = sc.flatten()
    assert len(f["prims"]) == 62976 + 2 and f["vertex_normals"] is not None
    img = sc.integrator().render(sc, seed=12, spp=4)
    ref, _ = oracle_render(ob, sc, 12, 4)
    assert np.array_equal(img, ref) and img.mean() > 0.05
