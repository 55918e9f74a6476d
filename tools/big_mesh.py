"""SURVEY f-4 stress case: a 64 800-triangle mesh (no LDS image: BVH through the vector caches), 512^2 x 32 spp"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
nu = nv = int(sys.argv[1]) if len(sys.argv) > 1 else 180
th = np.linspace(0, np.pi, nv + 1)[:, None]; ph = np.linspace(0, 2 * np.pi, nu, endpoint=False)[None, :]
r = 1.0 + 0.08 * np.sin(9 * th) * np.cos(7 * ph)                      # a bumpy ball
P = np.stack([r * np.sin(th) * np.cos(ph), r * np.cos(th) * np.ones_like(ph), r * np.sin(th) * np.sin(ph)], axis=-1)
idx = lambda i, j: i * nu + (j % nu) + 1
path = os.path.join(tempfile.mkdtemp(), "ball.obj")
with open(path, "w") as f:
    for p in P.reshape(-1, 3): f.write(f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}\n")
    for i in range(nv):
        for j in range(nu):
            f.write(f"f {idx(i, j)} {idx(i + 1, j + 1)} {idx(i + 1, j)}\nf {idx(i, j)} {idx(i, j + 1)} {idx(i + 1, j + 1)}\n")
T = mi.ScalarTransform4f
sc = mi.load_dict({
    "type": "scene", "integrator": {"type": "path", "max_depth": 6},
    "sensor": {"type": "perspective", "fov": 45, "to_world": T().look_at([0, 1, 5], [0, 0, 0], [0, 1, 0]),
               "film": {"type": "hdrfilm", "width": 512, "height": 512, "rfilter": {"type": "tent"}},
               "sampler": {"type": "independent", "sample_count": 32}},
    "ball": {"type": "obj", "filename": path, "merge_quads": False, "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.7, 0.5, 0.4]}}},
    "floor": {"type": "rectangle", "to_world": T().translate([0, -1.2, 0]) @ T().rotate([1, 0, 0], -90) @ T().scale([5, 5, 1]), "bsdf": {"type": "diffuse"}},
    "light": {"type": "rectangle", "to_world": T().translate([0, 4, 0]) @ T().rotate([1, 0, 0], 90), "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [10, 10, 10]}}, "bsdf": {"type": "diffuse"}}})
n = len(sc.flatten()["prims"])
for _ in range(3):
    img = mi.render(sc, seed=0); st = mi.default_context().stats()
print(f"{n} primitives, 512 x 512 x 32 spp: kernel {st['kernel_ms']:.1f} ms = {st['samples'] / st['kernel_ms'] / 1e3:.0f} Msamples/s, "
      f"segments/sample {st['segments'] / st['samples']:.2f}, mean {img.mean():.4f}", flush=True)
