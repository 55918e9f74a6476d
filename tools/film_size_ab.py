"""Same Cornell box, same number of samples, different film shapes: is the rate per traced path the same?  (config 5's bands against config 2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
S = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/scenes/cbox.xml")
ctx = mi.default_context()
def run(label, res, spp, crop=None, reps=3, **kw):
    sc = mi.load_file(S, res=res, spp=spp)
    integ = sc.integrator()
    best = None
    for _ in range(reps):
        integ.render(sc, seed=0, spp=spp, crop=crop, **kw)
        st = ctx.stats()
        if best is None or st["bounce_ms"] < best["bounce_ms"]: best = st
    n = best["samples"]
    live = [int(x) for x in best["live"][:6]]
    print(f"{label:38s} traced {n/1e6:7.1f} M paths  passes {best['passes']:2d} launches {best['bounce_launches']:3d}  bounce {best['bounce_ms']:8.3f} ms = {best['bounce_ms']*1e9/n:6.1f} ps/path  "
          f"kernel {best['kernel_ms']:8.3f} ms  segs/path {best['segments']/n:.3f}  live/path {[round(x/n,3) for x in live]}", flush=True)
run("512^2 x 256", 512, 256)
run("4096^2 x 4 (full film)", 4096, 4)
run("4096^2 x 8 (full film)", 4096, 8)
run("4096 x 64 band (rows 0-63) x 1024", 4096, 1024, crop=(0, 0, 4096, 64))
run("4096 x 64 band (rows 2048-2111) x 1024", 4096, 1024, crop=(0, 2048, 4096, 64))
run("4096 x 64 band (rows 2048-2111) x 256", 4096, 256, crop=(0, 2048, 4096, 64))
run("4096 x 512 band (rows 1792-2303) x 128", 4096, 128, crop=(0, 1792, 4096, 512))
