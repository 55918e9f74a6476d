"""kernel time of the headline render against the number of paths kept in flight per pass"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
sc = mi.load_file("tests/scenes/cbox.xml", res=512, spp=256)
ref = None
for mib in (2, 4, 8, 16, 32, 64):
    best = 1e9
    for i in range(3):
        img = sc.integrator().render(sc, seed=0, spp=256, pass_paths=mib << 20)
        st = mi.default_context().stats(); best = min(best, st["kernel_ms"])
    if ref is None: ref = img
    print(f"pass_paths {mib:3d} Mi: kernel {best:.2f} ms, bounce {st['bounce_ms']:.2f} ms, passes {st['passes']}, launches {st['bounce_launches']}, same image {np.array_equal(img, ref)}", flush=True)
