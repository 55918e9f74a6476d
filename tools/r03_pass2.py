"""ring scene, 1024^2 x 64 spp: kernel time by paths per pass (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd as mi
sc = mi.load_file("tests/scenes/testring.xml", res=1024, spp=512)
integ = sc.integrator()
for mi_paths in (64, 128, 256, 0):
    best = 1e9
    for i in range(2):
        integ.render(sc, seed=0, spp=512, pass_paths=mi_paths << 20); st = mi.default_context().stats(); best = min(best, st["kernel_ms"])
    print(f"{mi_paths:3d} Mi paths per pass: {best:7.2f} ms  passes {st['passes']}", flush=True)
