"""Lane efficiency of the BVH traversal by depth (library built with -DPBRT_BVH_PROBE; PBRT_HIP_LIB points at it): per max_depth
budget the wave trips x 64 against the lane trips of the node walk (phase 1) and of the primitive tests (phase 2), closest-hit and
any-hit queries apart.  Differences between consecutive budgets are the bounces.  usage: python tools/bvh_probe.py [scene.xml res spp]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
capi = __import__("importlib").import_module("physics-based-ray-tracing_amd._capi")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = sys.argv[1] if len(sys.argv) > 1 else "tests/scenes/testring.xml"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lib = capi.load_library()
buf = (C.c_ulonglong * 8)()
prev = np.zeros(8)
for md in (1, 2, 3, 4, 6):
    sc = mi.load_file(os.path.join(root, scene), res=res, spp=spp, max_depth=md)
    lib.pbrt_debug_bvh_probe(None, 1)
    sc.integrator().render(sc, seed=0, spp=spp)
    st = mi.default_context().stats()
    assert lib.pbrt_debug_bvh_probe(buf, 0) == 0
    v = np.array(list(buf), dtype=np.float64)
    dlt = v - prev
    prev = v
    def eff(a, b): return f"{b / a * 100:5.1f} %" if a else "   -   "
    print(f"max_depth {md}: live {list(st['live'][:md])}\n   this budget adds: closest hit: node walk {dlt[0]/1e6:9.1f} M lane-slots, {eff(dlt[0], dlt[1])} used | "
          f"prim tests {dlt[2]/1e6:9.1f} M, {eff(dlt[2], dlt[3])} used || any hit: node walk {dlt[4]/1e6:9.1f} M, {eff(dlt[4], dlt[5])} | prim tests {dlt[6]/1e6:9.1f} M, {eff(dlt[6], dlt[7])}", flush=True)
