"""BASELINE configs 3 and 4 at full size on one GPU (config 2 is bench.py): prints the library's own statistics"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
S = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "scenes")
us = mi.load_file(os.path.join(S, "us_sphere_box.xml"))
ui = us.integrator()
ppr = 838912                      # 5 x 64 x 838912 = 268 451 840 paths ("1024 x 1024 x 256 spp")
for i in range(2):
    t = time.perf_counter(); ui._acquire(us, ui.quirks, paths_per_ray=ppr); wall = time.perf_counter() - t
    st = mi.default_context().stats()
print(f"config 3 us_sphere_box 5x64x{ppr}: kernel {st['kernel_ms']:.1f} ms (wall {wall*1e3:.1f}) = {st['samples']/st['kernel_ms']/1e3:.0f} Mpaths/s, "
      f"segments/path {st['segments']/st['samples']:.3f}, passes {st['passes']}, model GB {st['model_bytes']/1e9:.2f}", flush=True)
sc = mi.load_file(os.path.join(S, "testring.xml"), res=1024, spp=512)
for i in range(2):
    t = time.perf_counter(); mi.render(sc, seed=0); wall = time.perf_counter() - t
    st = mi.default_context().stats()
print(f"config 4 testring 1024x1024x512: kernel {st['kernel_ms']:.1f} ms (wall {wall*1e3:.1f}) = {st['samples']/st['kernel_ms']/1e3:.0f} Msamples/s, "
      f"segments/sample {st['segments']/st['samples']:.3f}, passes {st['passes']}", flush=True)
