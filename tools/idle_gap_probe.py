"""Does a kernel that uses scratch memory pay for a host-side pause before it?  (The fused emitter-ray instance of k_us_bounce -- 17
spilled VGPRs -- ran its 17 launches per step 1.2 ms apart inside `python bench.py`, after the CPU legs of the config before it.)
Acquisitions back to back, then after a pause of PAUSE seconds, kernel ms of each (run on the GPU box):
    PBRT_US_EMIT_FUSED=1 python tools/idle_gap_probe.py [pause_s]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pbrt_amd as mi
pause = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
ctx = mi.default_context()
sc = mi.load_file(os.path.join(ROOT, "tests/scenes/us_sphere_box.xml"), primary_rays="emitter")
ui = sc.integrator()
def runs(n):
    out = []
    for _ in range(n):
        ui._acquire(sc, ui.quirks, paths_per_ray=838912, seed=0)
        st = ctx.stats()
        out.append((round(st["kernel_ms"], 2), round(st["bounce_ms"], 2)))
    return out
print("fused" if os.environ.get("PBRT_US_EMIT_FUSED") == "1" else "two-kernel", "warm:", runs(4), flush=True)
time.sleep(pause)
print(f"after {pause:g} s idle:", runs(4), flush=True)
import numpy as np
t0 = time.time()
while time.time() - t0 < pause:      # a busy host instead of an idle one
    np.linalg.svd(np.random.rand(300, 300))
print(f"after {pause:g} s of host work:", runs(4), flush=True)
