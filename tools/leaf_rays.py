"""closest-hit leaf operator on random rays inside a scene's bounds (for kernel-trace comparisons of BVH cost)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
n = 1 << 22
rng = np.random.default_rng(0)
for path, kw, lo, hi in (("tests/scenes/testring.xml", dict(res=8), -0.1, 0.1), ("tests/scenes/us_cone_box.xml", {}, -0.14, 0.14)):
    sc = mi.load_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path), **kw)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    if "cone" in path: o[:, 2] = rng.uniform(0.0, 0.3, n)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    r = sc.ray_intersect(o, d)
    print(path, "prims", len(sc.flatten()["prims"]), "hit fraction", r["valid"].mean(), flush=True)
    r = sc.ray_test(o, d)
