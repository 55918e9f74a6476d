"""two independent renders on two contexts (two HIP streams) at once vs one after the other: does the GPU have
idle capacity that more concurrency could use?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
from pbrt_amd import _capi
S = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "scenes", "cbox.xml")
ctxs = [_capi.Context(0), _capi.Context(0)]
scenes = []
for cx in ctxs:
    sc = mi.load_file(S, res=512, spp=256)
    f = sc.flatten()
    sc._dev = _capi.DeviceScene(cx, f["prims"], f["materials"], f["emitters"], f["light_prims"], f["light_cdf"], sc.accel)
    scenes.append(sc)
def render(sc, out):
    out.append(mi.render(sc, seed=0))
for sc in scenes:
    render(sc, [])
t = time.perf_counter()
for _ in range(5):
    for sc in scenes: render(sc, [])
seq = (time.perf_counter() - t) / 5
t = time.perf_counter()
for _ in range(5):
    th = [threading.Thread(target=render, args=(sc, [])) for sc in scenes]
    [x.start() for x in th]; [x.join() for x in th]
par = (time.perf_counter() - t) / 5
print(f"two renders sequential {seq*1e3:.2f} ms, concurrent {par*1e3:.2f} ms", flush=True)
