"""Pins the oracle's analytic cone primitive (PBRT_PRIM_CONE, DESIGN.md D8: 'cone' of MitsubaScenes/Cone_Box.xml =
closed unit cone under to_world) with answers it did not produce itself: hand-computed rays on the unit cone, an
f64 closed-form solver on the phantom's own transform, and a fine tessellation of the same surface."""
import numpy as np
import pytest

from conftest import scene_path


def _scene(mi, **cone):
    return mi.load_dict({"type": "scene", "c": {"type": "cone", **cone}})


def _isect(ob, sc, o, d, accel=None):
    o, d = np.asarray(o, np.float32), np.asarray(d, np.float32)
    osc = ob.OracleScene.from_scene(sc) if accel is None else ob.OracleScene.from_scene(sc, accel)
    return osc.ray_intersect(o, d, np.full(len(o), np.inf, np.float32))


def test_unit_cone_known_rays(mi, ob):
    sc = _scene(mi)
    P = sc.flatten()["prims"]
    assert len(P) == 1 and P["type"][0] == 3
    assert np.allclose(P["g"][0].reshape(3, 4), np.eye(4)[:3])              # world -> object of the identity
    s = np.float32(1 / np.sqrt(2))
    o = [[0, 0, -1], [2, 0, 0.5], [0, 0, 3], [0, 0, 0.5], [2, 0, 2], [2, 0, 0.2], [0.5, 0, -0.5], [0.25, 0.25, 0.25], [0, 3, 0.5]]
    d = [[0, 0, 1], [-1, 0, 0], [0, 0, -1], [1, 0, 0], [-1, 0, 0], [-s, 0, s], [-s, 0, s], [0, 0, -1], [0, 1, 0]]
    t, prim, u, v = _isect(ob, sc, o, d)
    #        base    side   apex  inside->side  mirror nappe: miss  parallel outside: miss  parallel -> base  inside->base  away
    want = [1.0, 1.5, 2.0, 0.5, np.inf, np.inf, 0.5 * np.sqrt(2), 0.25, np.inf]
    assert np.allclose(t, want, rtol=2e-6)
    assert np.array_equal(prim != 0xFFFFFFFF, np.isfinite(want))
    assert list(u[[0, 1, 3, 6, 7]]) == [1, 0, 0, 1, 1] and not v.any()     # u: 1 = base disc, 0 = lateral surface
    # tmax cuts the candidates like every other primitive
    osc = ob.OracleScene.from_scene(sc)
    o2, d2 = np.array([[2, 0, 0.5]] * 3, np.float32), np.array([[-1, 0, 0]] * 3, np.float32)
    t2, p2, *_ = osc.ray_intersect(o2, d2, np.array([1.4, 1.5, 1.6], np.float32))
    assert list(p2 != 0xFFFFFFFF) == [False, True, True]
    assert list(osc.ray_test(o2, d2, np.array([1.4, 1.5, 1.6], np.float32))) == [False, True, True]


def _closed_form(M, o, d):
    """f64: nearest t >= 0 of the ray with the closed unit cone under the 4x4 to_world M (inf = miss)"""
    Wi = np.linalg.inv(M)
    oo = o @ Wi[:3, :3].T + Wi[:3, 3]
    dd = d @ Wi[:3, :3].T
    best = np.full(len(o), np.inf)
    a = dd[:, 0] ** 2 + dd[:, 1] ** 2 - dd[:, 2] ** 2
    b = 2 * (oo[:, 0] * dd[:, 0] + oo[:, 1] * dd[:, 1] + (1 - oo[:, 2]) * dd[:, 2])
    c = oo[:, 0] ** 2 + oo[:, 1] ** 2 - (1 - oo[:, 2]) ** 2
    disc = b * b - 4 * a * c
    with np.errstate(all="ignore"):
        for sgn in (-1.0, 1.0):
            t = (-b + sgn * np.sqrt(disc)) / (2 * a)
            z = oo[:, 2] + t * dd[:, 2]
            ok = (disc >= 0) & (t >= 0) & (z >= 0) & (z <= 1)
            best = np.where(ok & (t < best), t, best)
        t = -oo[:, 2] / dd[:, 2]
        x, y = oo[:, 0] + t * dd[:, 0], oo[:, 1] + t * dd[:, 1]
        ok = (t >= 0) & (x * x + y * y <= 1)
        best = np.where(ok & (t < best), t, best)
    return best


def test_phantom_cone_against_f64_closed_form_and_fine_mesh(mi, ob, capi):
    sc = mi.load_file(scene_path("us_cone_box.xml"))
    shapes = sc.shapes()
    cone_shape = [i for i, s in enumerate(shapes) if s.id() == "cone"][0]
    M = shapes[cone_shape].to_world.matrix
    P = sc.flatten()["prims"]
    assert len(P) == 6 and (P["type"] == 3).sum() == 1
    ci = int(np.nonzero(P["type"] == 3)[0][0])
    assert np.allclose(P["g"][ci].reshape(3, 4) @ M, np.eye(4)[:3], atol=1e-5)  # g = world -> object
    rng = np.random.default_rng(5)
    n = 40000
    o = rng.uniform(-0.14, 0.14, (n, 3)) + [0, 0, 0.1]
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    only = mi.load_dict({"type": "scene", "c": {"type": "cone", "to_world": shapes[cone_shape].to_world}})
    t, prim, u, v = _isect(ob, only, o32, d32)
    want = _closed_form(M, o32.astype(np.float64), d32.astype(np.float64))
    hit, whit = prim != 0xFFFFFFFF, np.isfinite(want)
    assert 0.03 < whit.mean() < 0.9
    assert (hit != whit).mean() < 2e-3                                       # grazing rays may flip in f32
    both = hit & whit
    rel = np.abs(t[both] - want[both]) / want[both]
    assert np.quantile(rel, 0.999) < 1e-4 and np.median(rel) < 1e-6
    # the same surface as a fine mesh (256 segments x 8 rings): hits agree to the tessellation error
    mesh = mi.load_dict({"type": "scene", "c": {"type": "cone", "to_world": shapes[cone_shape].to_world, "tessellate": True,
                                                 "segments": 256, "rings": 8}})
    tm, pm, *_ = _isect(ob, mesh, o32, d32)
    mhit = pm != 0xFFFFFFFF
    assert (mhit != hit).mean() < 0.01
    both = hit & mhit
    assert np.quantile(np.abs(t[both] - tm[both]), 0.99) < 2e-4              # scene scale 0.1: chord error of 256 segments
    # inside the full phantom the cone keeps its place among the walls, and brute force == BVH bit for bit
    A = _isect(ob, sc, o32, d32, capi.ACCEL_BRUTE)
    B = _isect(ob, sc, o32, d32, capi.ACCEL_BVH)
    for x, y in zip(A, B):
        assert np.array_equal(x, y)
    sel = A[1] == ci
    assert sel.sum() > 1000 and np.array_equal(A[0][sel], t[sel])


def test_cone_rejects_what_it_cannot_represent(mi, ob, capi):
    with pytest.raises(ValueError):
        _scene(mi, to_world=mi.ScalarTransform4f().scale([1, 1, 0]))        # singular
    with pytest.raises(NotImplementedError):
        _scene(mi, flip_normals=True).flatten()
    with pytest.raises(NotImplementedError):                                  # no area emitters on analytic cones
        _scene(mi, emitter={"type": "area", "radiance": {"type": "rgb", "value": [1, 1, 1]}}).flatten()
    sc = _scene(mi)
    desc = sc.flatten()
    bad = desc["prims"].copy()
    bad["g"][0] = 0                                                          # singular world -> object matrix
    with pytest.raises(Exception):
        ob.OracleScene(bad, desc["materials"], desc["emitters"], desc["light_prims"], desc["light_cdf"])


def test_cone_normals_through_direct_illumination(mi, ob):
    """A diffuse cone under a point light: L = rho / pi * I / r^2 * cos(theta) depends on the normal the oracle reports.
    Checked on the lateral surface (n = (1, 0, 1) / sqrt 2 at (0.5, 0, 0.5)), on the base disc (n = -z), and on a
    squashed, rotated cone where the normal is the inverse-transpose image of the object-space gradient."""
    from conftest import oracle_render
    rho, I = 0.6, 40.0

    def radiance(to_world, cam_from, look_at, light):
        d = {"type": "scene", "integrator": {"type": "direct"},
             "sensor": {"type": "perspective", "fov": 0.2, "near_clip": 1e-3, "far_clip": 100.0,
                        "to_world": mi.ScalarTransform4f().look_at(cam_from, look_at, [0, 1, 0]),
                        "sampler": {"type": "independent", "sample_count": 64},
                        "film": {"type": "hdrfilm", "width": 3, "height": 3, "rfilter": {"type": "box"}}},
             "c": {"type": "cone", "to_world": to_world, "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [rho] * 3}}},
             "bulb": {"type": "point", "position": light, "intensity": {"type": "rgb", "value": [I] * 3}}}
        img, _ = oracle_render(ob, mi.load_dict(d), seed=1, spp=64)
        return img[1, 1].mean()

    def expect(p, n, light):
        wl = np.asarray(light, float) - p
        r2 = wl @ wl
        return rho / np.pi * I / r2 * max(0.0, n @ wl / np.sqrt(r2))

    T = mi.ScalarTransform4f
    s = 1 / np.sqrt(2)
    p, n, light = np.array([0.5, 0, 0.5]), np.array([s, 0, s]), [3.0, 1.0, 2.0]
    assert radiance(T(), [4.5, 0.5, 2.5], list(p), light) == pytest.approx(expect(p, n, light), rel=2e-3)
    p, n, light = np.array([0.2, 0.1, 0.0]), np.array([0, 0, -1.0]), [1.0, 0.5, -2.0]
    assert radiance(T(), [0.5, 0.3, -4.0], list(p), light) == pytest.approx(expect(p, n, light), rel=2e-3)
    tw = T().translate([0.2, -0.1, 0.3]) @ T().rotate([0.3, 1, 0.2], 35) @ T().scale([0.5, 1.5, 2.0])
    M = tw.matrix
    q = np.array([0.6 * np.cos(0.7), 0.6 * np.sin(0.7), 0.4])                    # on the unit cone: r = 1 - z
    p = M[:3, :3] @ q + M[:3, 3]
    n = np.linalg.inv(M[:3, :3]).T @ np.array([q[0], q[1], 1 - q[2]])
    n /= np.linalg.norm(n)
    light = list(p + 2.0 * n + [0.3, -0.2, 0.1])
    cam = list(p + 5.0 * (n + [0.1, 0.2, -0.1]))
    assert radiance(tw, cam, list(p), light) == pytest.approx(expect(p, n, light), rel=3e-3)
