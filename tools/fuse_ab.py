"""A/B of the fuse plans of the brute-force bounce kernels on cbox 512^2 x 256 spp: which depths start a two-bounce launch
(include/pbrt_hip.h PBRT_FILM_FUSE_PLAN).  Prints kernel ms (best of 5), bounce ms, launches, and checks that the film
does not change.   usage: python tools/fuse_ab.py [plan ...]   (hex masks; default 0 1 4 5 15 3f)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
capi = __import__("importlib").import_module("physics-based-ray-tracing_amd._capi")
plans = [int(x, 16) for x in sys.argv[1:]] or [0x0, 0x1, 0x4, 0x5, 0x15, 0x3f]
sc = mi.load_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/scenes", os.environ.get("FUSE_SCENE", "cbox.xml")), res=512, spp=256)
integ = sc.integrator()
ctx = mi.default_context()
ref = None
for p in plans:
    best = None
    for _ in range(6):
        img = integ.render(sc, seed=0, spp=256, flags=capi.film_fuse_plan(p))
        st = ctx.stats()
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    if ref is None:
        ref = img
    print(f"plan {p:#04x}: kernel {best['kernel_ms']:.3f} ms  bounce {best['bounce_ms']:.3f} ms  launches {best['bounce_launches']}  "
          f"model GB {best['bounce_model_bytes']/1e9:.2f}  live {list(best['live'][:6])}  same film {bool(np.array_equal(img, ref))}  "
          f"-> {512*512*256/best['kernel_ms']/1e3:.0f} Msamples/s", flush=True)
