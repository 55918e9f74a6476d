#!/usr/bin/env python3
"""Radiance mode: the Cornell box of BASELINE config 2 through the Mitsuba-shaped API.
    python examples/render_cbox.py [res] [spp] [out.npy]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import pbrt_amd as mi

res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out = sys.argv[3] if len(sys.argv) > 3 else "cbox.npy"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = mi.load_file(os.path.join(root, "tests", "scenes", "cbox.xml"), res=res, spp=spp)   # same overrides as mi.load_file(..., res=, spp=)
mi.render(scene, seed=0)                                                                     # warm-up: upload, first launch
t = time.perf_counter()
img = mi.render(scene, seed=0)
dt = time.perf_counter() - t
st = mi.default_context().stats()
print(f"{res} x {res} x {spp} spp: {dt * 1e3:.2f} ms wall, {st['kernel_ms']:.2f} ms on the GPU = "
      f"{res * res * spp / st['kernel_ms'] / 1e3:.0f} Msamples/s; mean radiance {img.mean():.4f}")
np.save(out, img)
