"""A/B of k_bounce_pool (BVH scenes, bounces >= 1: closest hits first, shading in full waves) against k_bounce: kernel ms, same film.
usage: python tools/pool_ab.py [scene.xml res spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
capi = __import__("importlib").import_module("physics-based-ray-tracing_amd._capi")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = sys.argv[1] if len(sys.argv) > 1 else "tests/scenes/testring.xml"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
sc = mi.load_file(os.path.join(root, scene), res=res, spp=spp)
integ = sc.integrator(); ctx = mi.default_context()
ref = None
for name, fl in (("k_bounce_pool", 0), ("k_bounce", capi.FILM_NO_HIT_POOL), ("k_bounce_pool", 0), ("k_bounce", capi.FILM_NO_HIT_POOL)):
    best = None
    for _ in range(3):
        img = integ.render(sc, seed=0, spp=spp, flags=fl)
        st = ctx.stats()
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    if ref is None: ref = img
    print(f"{name:14s}: kernel {best['kernel_ms']:8.3f} ms bounce {best['bounce_ms']:8.3f} ms  segments {best['segments']} shadow {best['shadow_rays']} live {list(best['live'][:6])} same film {bool(np.array_equal(img, ref))}", flush=True)
