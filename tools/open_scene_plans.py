"""Fuse plans on open scenes (few paths survive a bounce): kernel time of pairs / triples / 4 + 2 / one launch per pass on two of the
random scenes of tests/test_gpu_random_scenes.py at 512^2 x 128 spp; same film for every plan."""
import os, sys, tempfile, pathlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import pbrt_amd as mi
from test_gpu_random_scenes import _random_scene
capi = __import__("importlib").import_module("physics-based-ray-tracing_amd._capi")
tmp = pathlib.Path(tempfile.mkdtemp())
for seed, shape in ((1000, (3, 4, 0, 0)), (1001, (5, 10, 12, 0))):
    sc = _random_scene(mi, tmp, seed, *shape)
    f = sc.sensors()[0].film(); f.width, f.height, f.crop = 512, 512, (0, 0, 512, 512)
    integ = sc.integrator(); integ.max_depth = 6
    ctx = mi.default_context()
    ref = None
    for plan in ("15", "1b", "17", "1f"):   # pairs, triples, 4 + 2, one launch
        best = None
        for _ in range(4):
            img = integ.render(sc, seed=0, spp=128, flags=capi.film_fuse_plan(int(plan, 16)))
            st = ctx.stats()
            if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
        if ref is None: ref = img
        print(f"seed {seed} ({len(sc.flatten()['prims'])} prims) plan {plan}: kernel {best['kernel_ms']:.3f} ms  live/path {[round(x/best['samples'],3) for x in best['live'][:6]]} same {bool(np.array_equal(img, ref))}", flush=True)
    best = None
    for _ in range(4):
        img = integ.render(sc, seed=0, spp=128)
        st = ctx.stats()
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    print(f"seed {seed} learnt plan: kernel {best['kernel_ms']:.3f} ms  launches per pass {best['bounce_launches'] // best['passes']} same {bool(np.array_equal(img, ref))}", flush=True)
