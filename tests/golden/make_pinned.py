"""Generates the fixtures K9 (ultrasound) and K10 (radiance) with tests/golden/ref_transcription.py, the float64
restatement written from the reference's Python -- NOT from oracle/oracle.cpp (see that module's header).

Run in the build container (reads the reference's cbox OBJ quads as data):   python tests/golden/make_pinned.py
Outputs, committed:  k9_us_plate.npz, k9_us_sphere_box.npz, k9_us_two_plates.npz, k9_us_two_plates_drjit.npz, k9_us_plate_box.npz,
                     k10_cbox_paths.npz, k11_meshes.npz, k12_emitter_sensor.npz

K9  scenes:  'plate'       the scene USMain.py:26-90 builds (tilted plate 5 cm ahead, back wall at 1 m; integrator block :28-42)
             'sphere_box'  MitsubaScenes/Sphere_Box.xml:2-101 with the author-intent transforms
             'two_plates'  a narrow tilted plate in front of a wall: the scene whose SECOND-bounce echoes reach the receive
                           elements (in both variants of the reference's loop: scalar and dr.while_loop)
    every (angle, element, path k) ray is traced by us_trace_single_ray with the draws rng4(ray, k, bounce, seed);
    the fixture holds every echo bin (index, summed pressure, summed envelope, the smallest decision margin among
    its contributors) and a set of single-bounce BSDF records (wi, n, sh_frame.s, s1, s2 -> wo, pdf, amplitude).
K10 the Cornell box of scenes/cbox.xml (camera :11-21, materials :36-54, spheres :115-129, quads from
    scenes/meshes/cbox_*.obj, luminaire translated by -0.01 :60-62, radiance (1,1,1) :75 / DESIGN D1), 24 x 24 pixels,
    samples 0 and 1 of every pixel, max_depth 6: per-sample radiance of path_radiance + decision margin.
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_transcription as rt  # noqa: E402

REF = "/root/reference"


def Rx(a):
    c, s = math.cos(math.radians(a)), math.sin(math.radians(a))
    M = np.eye(4)
    M[1, 1], M[1, 2], M[2, 1], M[2, 2] = c, -s, s, c
    return M


def Ry(a):
    c, s = math.cos(math.radians(a)), math.sin(math.radians(a))
    M = np.eye(4)
    M[0, 0], M[0, 2], M[2, 0], M[2, 2] = c, s, -s, c
    return M


def Tr(x, y, z):
    M = np.eye(4)
    M[:3, 3] = [x, y, z]
    return M


def Sc(x, y, z):
    return np.diag([x, y, z, 1.0])


US_SCENES = {
    # USMain.py:26-90: integrator :28-42, sensor look_at :53-57 (identity), flat_plate :67-75, wall_back :79-87
    "plate": dict(
        params=dict(max_depth=10, fs=50e6, frequency=5e6, sound_speed=1540.0, attenuation=0.2, main_beam_angle=24.0, cutoff_angle=30.0,
                    n_elements=64, pitch=1.2e-4, time_samples=10000, angles_deg=[-15.0, -7.5, 0.0, 7.5, 15.0]),
        look_at=([0, 0, 0], [0, 0, 0.03], [0, 1, 0]),
        shapes=[dict(type="rectangle", to_world=Tr(0, 0, 0.05) @ Ry(45) @ Sc(0.17, 0.17, 0.14), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0, 0, 1) @ Ry(180) @ Sc(0.05, 0.05, 1), impedance=7.8, roughness=0.7)],
        seed=11, ppr=2),
    # MitsubaScenes/Sphere_Box.xml: integrator :2-15, sensor :16-34 (identity), sphere :36-45, walls :47-101 (T @ R @ S)
    "sphere_box": dict(
        params=dict(max_depth=10, fs=50e6, frequency=3e6, sound_speed=1480.0, attenuation=0.1, main_beam_angle=24.0, cutoff_angle=30.0,
                    n_elements=64, pitch=1.2e-4, time_samples=10000, angles_deg=[-15.0, -7.5, 0.0, 7.5, 15.0]),
        look_at=([0, 0, 0], [0, 0, 0.05], [0, 1, 0]),
        shapes=[dict(type="sphere", center=[0, 0, 0.08], radius=0.06, impedance=7.8, roughness=0.9),
                dict(type="rectangle", to_world=Tr(0, 0, 0.37) @ Ry(180) @ Sc(0.15, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(-0.15, 0, 0.12) @ Ry(90) @ Sc(0.25, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0.15, 0, 0.12) @ Ry(-90) @ Sc(0.25, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0, 0.15, 0.12) @ Rx(90) @ Sc(0.15, 0.25, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0, -0.15, 0.12) @ Rx(-90) @ Sc(0.15, 0.25, 1), impedance=7.8, roughness=0.7)],
        seed=7, ppr=3),
    # BASELINE config 3 read literally ("Sphere_Box.xml with CustomBSDF + CustomEmmitter"): the same phantom, every path's primary ray
    # drawn from CustomEmitter.sample_ray (CustomEmmitter.py:81-107) -- jitter inside the element, a steering angle from the a-th
    # fifth of [-15, 15] degrees, the element's steering delay, the cosine weight (PBRT_US_PRIMARY_EMITTER, include/pbrt_hip.h)
    "sphere_box_emitter": dict(
        params=dict(max_depth=10, fs=50e6, frequency=3e6, sound_speed=1480.0, attenuation=0.1, main_beam_angle=24.0, cutoff_angle=30.0,
                    n_elements=64, pitch=1.2e-4, time_samples=10000, angles_deg=[-15.0, -7.5, 0.0, 7.5, 15.0]),
        emitter=dict(number_of_elements=64, pitch=1.2e-4, element_width=1.0e-4, element_height=5.0e-4, radius=0.0, opening_angle=0.0,
                     number_of_rays_per_element=3, speed_of_sound=1480.0, steering_angle_min=-15.0, steering_angle_max=15.0),
        look_at=([0, 0, 0], [0, 0, 0.05], [0, 1, 0]),
        shapes=[dict(type="sphere", center=[0, 0, 0.08], radius=0.06, impedance=7.8, roughness=0.9),
                dict(type="rectangle", to_world=Tr(0, 0, 0.37) @ Ry(180) @ Sc(0.15, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(-0.15, 0, 0.12) @ Ry(90) @ Sc(0.25, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0.15, 0, 0.12) @ Ry(-90) @ Sc(0.25, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0, 0.15, 0.12) @ Rx(90) @ Sc(0.15, 0.25, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0, -0.15, 0.12) @ Rx(-90) @ Sc(0.15, 0.25, 1), impedance=7.8, roughness=0.7)],
        seed=17, ppr=3),
    # second-bounce ECHOES: in the two scenes above a path that goes on never deposits again (plate: it leaves the cut-off cone
    # or dies; sphere_box: it continues inside the sphere, from where no receive element is visible).  Here a narrow plate 2 cm
    # ahead, 7.5 mm off axis and tilted by 12 degrees, in front of a wall at 5 cm: under the reference's arithmetic the
    # "reflected" direction wi + 2 cos m, used as a world vector (quirks A5 / A9), heads on towards +z through the plate, hits the
    # wall, and most wall points see the receive element past the plate's edge -- so the continuation direction (:358-359), the
    # roulette division (:364-367) and the accumulated time of flight (:316) are in deposited VALUES
    "two_plates": dict(
        params=dict(max_depth=4, fs=50e6, frequency=3e6, sound_speed=1480.0, attenuation=0.1, main_beam_angle=24.0, cutoff_angle=30.0,
                    n_elements=64, pitch=1.2e-4, time_samples=10000, angles_deg=[-15.0, -7.5, 0.0, 7.5, 15.0]),
        look_at=([0, 0, 0], [0, 0, 0.05], [0, 1, 0]),
        shapes=[dict(type="rectangle", to_world=Tr(0.0075, 0, 0.02) @ Ry(180 + 12) @ Sc(0.003, 0.01, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0, 0, 0.05) @ Ry(180) @ Sc(0.05, 0.05, 1), impedance=7.8, roughness=0.5)],
        seed=3, ppr=4),
    # MitsubaScenes/Plate_Box.xml: integrator :2-15, plate :35-45, walls :46-100 (T @ R @ S): the tilted plate of Plane_Floating.xml
    # inside the box of Sphere_Box.xml.  Reflections off the 45-degree plate go sideways to the walls and on: the fixture of the set
    # with the longest chains (echoes deposited from the walls after two and three bounces)
    "plate_box": dict(
        params=dict(max_depth=10, fs=50e6, frequency=3e6, sound_speed=1480.0, attenuation=0.1, main_beam_angle=24.0, cutoff_angle=30.0,
                    n_elements=64, pitch=1.2e-4, time_samples=10000, angles_deg=[-15.0, -7.5, 0.0, 7.5, 15.0]),
        look_at=([0, 0, 0], [0, 0, 0.05], [0, 1, 0]),
        shapes=[dict(type="rectangle", to_world=Tr(0, 0, 0.05) @ Ry(45) @ Sc(0.17, 0.17, 0.02), impedance=7.8, roughness=0.9),
                dict(type="rectangle", to_world=Tr(0, 0, 0.37) @ Ry(180) @ Sc(0.15, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(-0.15, 0, 0.12) @ Ry(90) @ Sc(0.25, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0.15, 0, 0.12) @ Ry(-90) @ Sc(0.25, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0, 0.15, 0.12) @ Rx(90) @ Sc(0.15, 0.25, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0, -0.15, 0.12) @ Rx(-90) @ Sc(0.15, 0.25, 1), impedance=7.8, roughness=0.7)],
        seed=5, ppr=3),
    # the ultrasound twin of BASELINE config 4 (tests/scenes/us_testring.xml): the reference's TestRing/TestRing.obj (a data asset,
    # read here from tests/scenes/meshes/) with its vertex normals, axis across the beam, in the box of Sphere_Box.xml.  The mesh
    # case of the acquisition loop: Mesh::ray_intersect_triangle, interpolated shading normal, dp_du = p1 - p0
    "testring": dict(
        params=dict(max_depth=10, fs=50e6, frequency=3e6, sound_speed=1480.0, attenuation=0.1, main_beam_angle=24.0, cutoff_angle=30.0,
                    n_elements=64, pitch=1.2e-4, time_samples=10000, angles_deg=[-15.0, -7.5, 0.0, 7.5, 15.0]),
        look_at=([0, 0, 0], [0, 0, 0.05], [0, 1, 0]),
        shapes=[dict(type="obj", filename="meshes/TestRing.obj", to_world=Tr(0, 0.025, 0.09) @ Rx(90), impedance=7.8, roughness=0.9),
                dict(type="rectangle", to_world=Tr(0, 0, 0.37) @ Ry(180) @ Sc(0.15, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(-0.15, 0, 0.12) @ Ry(90) @ Sc(0.25, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0.15, 0, 0.12) @ Ry(-90) @ Sc(0.25, 0.15, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0, 0.15, 0.12) @ Rx(90) @ Sc(0.15, 0.25, 1), impedance=7.8, roughness=0.7),
                dict(type="rectangle", to_world=Tr(0, -0.15, 0.12) @ Rx(-90) @ Sc(0.15, 0.25, 1), impedance=7.8, roughness=0.7)],
        seed=13, ppr=3),
}


def read_obj(path):
    """v / vn / f (v//vn) of a triangle OBJ -> vertices [nv, 3], triangles [nt, 3], per-triangle vertex normals [nt, 3, 3]"""
    v, vn, tri, ntri = [], [], [], []
    for ln in open(path):
        t = ln.split()
        if not t:
            continue
        if t[0] == "v":
            v.append([float(x) for x in t[1:4]])
        elif t[0] == "vn":
            vn.append([float(x) for x in t[1:4]])
        elif t[0] == "f":
            ids = [tok.split("/") for tok in t[1:]]
            assert len(ids) == 3
            tri.append([int(i[0]) - 1 for i in ids])
            ntri.append([int(i[2]) - 1 for i in ids])
    return np.array(v), np.array(tri), np.array(vn)[np.array(ntri)]


def build_shapes(desc):
    out = []
    for d in desc:
        bs = dict(impedance=d["impedance"], roughness=d["roughness"])
        if d["type"] == "sphere":
            out.append(rt.Sphere(d["center"], d["radius"], bsdf=bs))
        elif d["type"] == "obj":   # Mitsuba `obj` shape under to_world: points by M, normals by the inverse transpose (a rotation here)
            v, tri, vn = read_obj(os.path.join(HERE, "..", "scenes", d["filename"]))
            M = np.asarray(d["to_world"], dtype=np.float64)
            assert np.allclose(M[:3, :3] @ M[:3, :3].T, np.eye(3))
            out.append(rt.TriMesh(v @ M[:3, :3].T + M[:3, 3], tri, bsdf=bs, vertex_normals=vn @ M[:3, :3].T))
        else:
            out.append(rt.Parallelogram.rectangle(d["to_world"], bsdf=bs))
    return out


EMIT_BLOCK = 0x80000000   # the RNG block of a path's emitter draws (include/pbrt_hip.h PBRT_US_PRIMARY_EMITTER)


def emitter_primary(S, a, e, ray, k):
    """the primary ray of path k of the (angle a, element e) pair when the scene names an emitter: CustomEmitter.sample_ray(0,
    (e + 1/2) / N, (u.x, u.y), (a + u.z) / n_angles) -- the acquisition grid stratifies the element pick and the steering range"""
    if "emitter" not in S:
        return None
    P = S["params"]
    u = rt.rng4(ray, k, EMIT_BLOCK, S["seed"])
    o, d, t, w, _ = rt.emitter_sample_ray(S["emitter"], 0.0, (e + 0.5) / P["n_elements"], (u[0], u[1]), (a + u[2]) / len(P["angles_deg"]))
    return o, d, t, w


def make_k9(name, S, variant="scalar"):
    shapes = build_shapes(S["shapes"])
    P, seed, ppr = S["params"], S["seed"], S["ppr"]
    T = rt.look_at(*S["look_at"])
    NA, NE = len(P["angles_deg"]), P["n_elements"]
    bins, recs = {}, []
    n_bounces = 0
    for a in range(NA):
        for e in range(NE):
            ray = a * NE + e
            for k in range(ppr):
                out = rt.us_trace_single_ray(shapes, T, P, a, e, lambda dep: rt.rng4(ray, k, dep, seed), variant,
                                             primary=emitter_primary(S, a, e, ray, k))
                n_bounces += len(out)
                for r in out:
                    recs.append(r)
                    if r["deposited"]:
                        b = bins.setdefault((a, r["recv"], r["t_idx"]), dict(p=0.0, env=0.0, env_abs=0.0, margin=math.inf, n=0))
                        b["p"] += r["pressure"]
                        b["env"] += r["envelope"]
                        b["env_abs"] += abs(r["envelope"])
                        b["margin"] = min(b["margin"], r["margin"])
                        b["n"] += 1
                # a decision that may flip in f32 changes everything after it: later bounces inherit the margin
                m = math.inf
                for r in out:
                    m = min(m, r["margin"])
                    if r["deposited"]:
                        b = bins[(a, r["recv"], r["t_idx"])]
                        b["margin"] = min(b["margin"], m)
    keys = sorted(bins)
    # single-bounce BSDF records: well-conditioned ones, spread over lobes / TIR / depths
    good = [r for r in recs if r["margin"] >= 1e-2]
    if "emitter" in S:
        # emitter rays meet the sphere at every incidence: micro-normals within 1.5 degrees of grazing (pdf = 1 / (4 |wi . m|) > 10)
        # pass the decision margin but leave f32 four digits of the pdf; the records are for well-conditioned samples
        good = [r for r in good if r["pdf"] <= 10.0]
    pick = []
    for cond in (lambda r: r["tir"], lambda r: r["reflect"] and not r["tir"], lambda r: not r["reflect"], lambda r: r["depth"] >= 1):
        sel = [r for r in good if cond(r)]
        pick += sel[:: max(1, len(sel) // 16)][:16]
    tag = name if variant == "scalar" else f"{name}_{variant}"
    meta = dict(scene=name, variant=variant, params=P, look_at=S["look_at"], seed=seed, paths_per_ray=ppr, n_bounces=n_bounces,
                **({"emitter": S["emitter"]} if "emitter" in S else {}),
                shapes=[{k: (np.asarray(v).tolist() if k in ("to_world", "center") else v) for k, v in d.items()} for d in S["shapes"]],
                depth_histogram={str(d): sum(1 for r in recs if r["depth"] == d) for d in sorted({r["depth"] for r in recs})},
                deposited_by_depth={str(d): sum(1 for r in recs if r["depth"] == d and r["deposited"]) for d in sorted({r["depth"] for r in recs})})
    np.savez_compressed(
        os.path.join(HERE, f"k9_us_{tag}.npz"), meta=json.dumps(meta),
        bin_index=np.array(keys, np.int32).reshape(-1, 3), bin_pressure=np.array([bins[k]["p"] for k in keys]),
        bin_envelope=np.array([bins[k]["env"] for k in keys]), bin_envelope_abs=np.array([bins[k]["env_abs"] for k in keys]),
        bin_margin=np.array([bins[k]["margin"] for k in keys]), bin_count=np.array([bins[k]["n"] for k in keys], np.int32),
        rec_wi=np.array([r["wi"] for r in pick]), rec_n=np.array([r["n"] for r in pick]), rec_sh_n=np.array([r["sh_n"] for r in pick]), rec_sh_s=np.array([r["sh_s"] for r in pick]),
        rec_s1=np.array([r["s1"] for r in pick]), rec_s2=np.array([r["s2"] for r in pick]), rec_wo=np.array([r["wo"] for r in pick]),
        rec_pdf=np.array([r["pdf"] for r in pick]), rec_a_resp=np.array([r["a_resp"] for r in pick]),
        rec_reflect=np.array([r["reflect"] for r in pick]), rec_tir=np.array([r["tir"] for r in pick]),
        rec_new_dir=np.array([r["new_dir"] for r in pick]), rec_shape=np.array([r["shape"] for r in pick], np.int32),
        rec_depth=np.array([r["depth"] for r in pick], np.int32))
    safe = sum(1 for k in keys if bins[k]["margin"] >= 1e-2)
    print(f"k9_us_{tag}.npz: {n_bounces} bounces {meta['depth_histogram']}, {len(keys)} bins ({safe} with margin >= 1e-2), {len(pick)} BSDF records")


# ---- K10 --------------------------------------------------------------------------------------------------------------
def obj_quad(path):
    v = [list(map(float, ln.split()[1:4])) for ln in open(path) if ln.startswith("v ")]
    f = [ln.split()[1:] for ln in open(path) if ln.startswith("f ")]
    assert len(f) == 1 and len(f[0]) == 4
    return [np.array(v[int(t.split("/")[0]) - 1]) for t in f[0]]


def cbox_shapes():
    white = dict(type="diffuse", reflectance=[0.885809, 0.698859, 0.666422])   # scenes/cbox.xml:36-42
    green = dict(type="diffuse", reflectance=[0.105421, 0.37798, 0.076425])    # :44-46
    red = dict(type="diffuse", reflectance=[0.570068, 0.0430135, 0.0443706])   # :48-50
    mesh = os.path.join(REF, "scenes", "meshes")

    def quad(name, bsdf, emitter=None, dy=0.0):
        a, b, c, d = (p + np.array([0.0, dy, 0.0]) for p in obj_quad(os.path.join(mesh, name)))
        assert np.allclose(a + c, b + d)
        return rt.Parallelogram(a, b - a, d - a, bsdf=bsdf, emitter=emitter)

    light = quad("cbox_luminaire.obj", white, dict(radiance=[1.0, 1.0, 1.0]), dy=-0.01)   # :58-86
    shapes = [light, quad("cbox_floor.obj", white), quad("cbox_ceiling.obj", white), quad("cbox_back.obj", white),
              quad("cbox_greenwall.obj", green), quad("cbox_redwall.obj", red),
              rt.Sphere([-0.3, -0.5, 0.2], 0.5, bsdf=dict(type="conductor")),                    # :115-121
              rt.Sphere([0.5, -0.75, -0.2], 0.25, bsdf=dict(type="dielectric", eta=1.5046 / 1.000277))]   # :123-129, bk7 / air
    return shapes, [light]


def make_k10():
    res, seed, max_depth, rr_depth, n_s = 24, 5, 6, 5, 2
    shapes, lights = cbox_shapes()
    cam = dict(x_fov=39.3077, width=res, height=res, near=0.001, far=100.0, to_world=rt.look_at([0, 0, 4], [0, 0, 0], [0, 1, 0]))
    L = np.zeros((n_s, res, res, 3))
    M = np.zeros((n_s, res, res))
    for s in range(n_s):
        for y in range(res):
            for x in range(res):
                pix = y * res + x
                uj = rt.rng4(pix, s, 0, seed)                       # block 0: pixel jitter
                o, d, tmax = rt.perspective_ray(cam, (x + uj[0]) / res, (y + uj[1]) / res)
                L[s, y, x], M[s, y, x] = rt.path_radiance(shapes, lights, o, d, tmax, (pix, s), seed, max_depth, rr_depth)
    meta = dict(res=res, seed=seed, max_depth=max_depth, rr_depth=rr_depth, samples=n_s, x_fov=39.3077, near=0.001, far=100.0,
                quads=[dict(p0=q.p0.tolist(), e1=q.e1.tolist(), e2=q.e2.tolist()) for q in shapes[:6]])
    np.savez_compressed(os.path.join(HERE, "k10_cbox_paths.npz"), meta=json.dumps(meta), radiance=L, margin=M)
    print(f"k10_cbox_paths.npz: {n_s} x {res} x {res} paths, mean radiance {L.mean():.5f}, "
          f"{(M >= 1e-3).mean() * 100:.1f} % with margin >= 1e-3")


# ---- K11: triangle meshes ---------------------------------------------------------------------------------------------
def ascii_ply(path):
    lines = open(path).read().split("\n")
    nv = int([l for l in lines if l.startswith("element vertex")][0].split()[2])
    nf = int([l for l in lines if l.startswith("element face")][0].split()[2])
    props = [l.split()[2] for l in lines[:lines.index("end_header")] if l.startswith("property") and "list" not in l]
    body = lines[lines.index("end_header") + 1:]
    v = np.array([[float(x) for x in body[i].split()] for i in range(nv)])[:, [props.index("x"), props.index("y"), props.index("z")]]
    t = []
    for i in range(nv, nv + nf):
        idx = [int(x) for x in body[i].split()]
        for k in range(2, idx[0]):
            t.append((idx[1], idx[k], idx[k + 1]))
    return v, np.array(t)


def make_k11():
    """(a) BASELINE config 1, the reference's own CPU-runnable case: scenes/simple.xml -- `direct` integrator (:5), perspective
    sensor with the default 50 mm focal length looking from (0,-12,5) at (0,0,1.25), up z (:7-12), box filter (:17), teapot.ply
    with diffuse (0.9, 0.9, 0) (:23-28), two `point` emitters of intensity 100 (:30-38); 64 x 64, samples 0..3 of every pixel.
    (b) interpolated shading normals: a 16-triangle ball with exact radial vertex normals under one point light."""
    here_meshes = os.path.join(os.path.dirname(HERE), "scenes", "meshes")
    v, t = ascii_ply(os.path.join(here_meshes, "teapot.ply"))          # byte-identical copy of scenes/meshes/teapot.ply
    teapot = rt.TriMesh(v, t, bsdf=dict(type="diffuse", reflectance=[0.9, 0.9, 0.0]))
    lights = [rt.PointLight([3, -10, 6], [100, 100, 100]), rt.PointLight([-3, -10, -2], [100, 100, 100])]
    res, seed, n_s = 64, 0, 4
    diag = 2.0 * math.atan(math.sqrt(36.0 ** 2 + 24.0 ** 2) / (2.0 * 50.0))     # 50 mm on 36 x 24 mm film, diagonal axis
    x_fov = math.degrees(2.0 * math.atan(math.tan(diag / 2.0) / math.sqrt(2.0)))   # aspect 1: width = diagonal / sqrt(2)
    cam = dict(x_fov=x_fov, width=res, height=res, near=1e-2, far=1e4, to_world=rt.look_at([0, -12, 5], [0, 0, 1.25], [0, 0, 1]))
    L = np.zeros((n_s, res, res, 3))
    for s_ in range(n_s):
        for y in range(res):
            for x in range(res):
                pix = y * res + x
                uj = rt.rng4(pix, s_, 0, seed)
                o, d, tmax = rt.perspective_ray(cam, (x + uj[0]) / res, (y + uj[1]) / res)
                L[s_, y, x], _ = rt.path_radiance([teapot], lights, o, d, tmax, (pix, s_), seed, 2, 5)
    # (b)
    import importlib.util
    spec = importlib.util.spec_from_file_location("mesh_util", os.path.join(os.path.dirname(HERE), "mesh_util.py"))
    mu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mu)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ball.obj")
        mu.write_uv_sphere_obj(path, n_lat=3, n_lon=4, normals=True)
        vb = np.array([[float(x) for x in ln.split()[1:4]] for ln in open(path) if ln.startswith("v ")])
        vn = np.array([[float(x) for x in ln.split()[1:4]] for ln in open(path) if ln.startswith("vn ")])
        fb = np.array([[int(tok.split("/")[0]) - 1 for tok in ln.split()[1:4]] for ln in open(path) if ln.startswith("f ")])
    ball = rt.TriMesh(vb, fb, bsdf=dict(type="diffuse", reflectance=[0.8, 0.7, 0.6]), vertex_normals=vn[fb])
    resb = 24
    camb = dict(x_fov=30.0, width=resb, height=resb, near=0.1, far=50.0, to_world=rt.look_at([0, 0, 5], [0, 0, 0], [0, 1, 0]))
    lb = [rt.PointLight([3, 4, 6], [60, 60, 60])]
    Lb = np.zeros((2, resb, resb, 3))
    for s_ in range(2):
        for y in range(resb):
            for x in range(resb):
                pix = y * resb + x
                uj = rt.rng4(pix, s_, 0, 4)
                o, d, tmax = rt.perspective_ray(camb, (x + uj[0]) / resb, (y + uj[1]) / resb)
                Lb[s_, y, x], _ = rt.path_radiance([ball], lb, o, d, tmax, (pix, s_), 4, 3, 5)
    meta = dict(simple=dict(res=res, seed=seed, samples=n_s, x_fov=x_fov), ball=dict(res=resb, seed=4, samples=2, n_lat=3, n_lon=4, max_depth=3))
    np.savez_compressed(os.path.join(HERE, "k11_meshes.npz"), meta=json.dumps(meta), simple=L.astype(np.float32), ball=Lb)
    print(f"k11_meshes.npz: simple.xml {n_s} x {res} x {res} samples, mean {L.mean():.5f}, lit {(L.sum(axis=3) > 0).mean() * 100:.1f} %; "
          f"ball mean {Lb.mean():.5f}, lit {(Lb.sum(axis=3) > 0).mean() * 100:.1f} %")


def make_k12():
    """K12: CustomEmitter.sample_position / sample_ray (CustomEmmitter.py:30-107; linear array with the source's defaults, and the
    convex branch :41-47) and UltraSensor.sample_ray (linear and convex, both direction branches) on seeded draws."""
    rng = np.random.default_rng(12)
    n = 512
    out = {}
    em_cases = {"linear": dict(number_of_elements=64, pitch=0.0003, element_width=0.0003, element_height=0.0005, radius=0.0,
                               opening_angle=0.0, number_of_rays_per_element=1, speed_of_sound=1540.0, steering_angle_min=-10.0,
                               steering_angle_max=10.0),
                "convex": dict(number_of_elements=48, pitch=0.0003, element_width=0.0003, element_height=0.0005, radius=0.05,
                               opening_angle=60.0, number_of_rays_per_element=4, speed_of_sound=1480.0, steering_angle_min=-15.0,
                               steering_angle_max=5.0)}
    draws = dict(time=rng.random(n) * 1e-6, s1=rng.random(n), s2=rng.random((n, 2)), s3=rng.random(n), wl=rng.random(n),
                 pos=rng.random((n, 2)), ap=rng.random((n, 2)))
    draws = {k: v.astype(np.float32).astype(np.float64) for k, v in draws.items()}   # the implementations take float32 inputs
    for name, P in em_cases.items():
        rec = np.array([np.concatenate([o, d, [t, w, pdf]]) for o, d, t, w, pdf in
                        (rt.emitter_sample_ray(P, draws["time"][i], draws["s1"][i], draws["s2"][i], draws["s3"][i]) for i in range(n))])
        out[f"emitter_{name}"] = rec
    look = ([0.0, 0.0, 0.0], [0.1, 0.0, 1.0], [0.0, 1.0, 0.0])
    T = rt.look_at(*look)
    sn_cases = {"linear": dict(num_elements_lateral=64, element_width=0.003, element_height=0.01, pitch=3e-4, radius=math.inf,
                               center_frequency=5e6, sound_speed=1540.0, directivity=1.0),
                "convex": dict(num_elements_lateral=32, element_width=0.003, element_height=0.01, pitch=3e-4, radius=0.04,
                               center_frequency=3e6, sound_speed=1540.0, directivity=0.7)}
    for name, P in sn_cases.items():
        for hemi in (1, 0):
            rec = np.array([np.concatenate([o, d, [w]]) for o, d, w in
                            (rt.ultra_sensor_sample_ray(P, T, draws["time"][i], draws["wl"][i], draws["pos"][i], draws["ap"][i], bool(hemi))
                             for i in range(n))])
            out[f"sensor_{name}_{hemi}"] = rec
    meta = dict(emitters=em_cases, sensors={k: {kk: (None if isinstance(vv, float) and math.isinf(vv) else vv) for kk, vv in v.items()}
                                            for k, v in sn_cases.items()}, look_at=look, n=n)
    np.savez_compressed(os.path.join(HERE, "k12_emitter_sensor.npz"), meta=json.dumps(meta), **draws, **out)
    print(f"k12_emitter_sensor.npz: {n} draws x {len(out)} cases")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "k12":
        make_k12()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "k9":   # one ultrasound fixture: python make_pinned.py k9 plate_box
        make_k9(sys.argv[2], US_SCENES[sys.argv[2]])
        sys.exit(0)
    for name, S in US_SCENES.items():
        make_k9(name, S)
    make_k9("two_plates", US_SCENES["two_plates"], "drjit")      # simulate_acquisition (CustomIntegrator.py:60-232)
    make_k10()
    make_k11()
    make_k12()
