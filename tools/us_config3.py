"""BASELINE config 3 only (5 x 64 x 838 912 paths on us_sphere_box): python tools/us_config3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd as mi
us = mi.load_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "scenes", "us_sphere_box.xml"))
ui = us.integrator()
best = 1e9
for i in range(4):
    ui._acquire(us, ui.quirks, paths_per_ray=838912)
    st = mi.default_context().stats()
    best = min(best, st["kernel_ms"])
print(f"config 3: best kernel {best:.2f} ms = {st['samples'] / best / 1e3:.0f} Mpaths/s, passes {st['passes']}", flush=True)
