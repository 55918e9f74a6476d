"""cost of each 64-row band of the cbox film (weak-scaling balance at N = 8: one band per rank)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pbrt_amd as mi
par = __import__("importlib").import_module("physics-based-ray-tracing_amd.parallel")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = mi.load_file("tests/scenes/cbox.xml", res=512, spp=256)
sc.device()
ctx = mi.default_context()
world = 512 // rows
ms = []
for r in range(world):
    for _ in range(2):
        par.render_tiles(sc, 256, 0, r, world, rows, device=torch.device("cuda", 0))
        st = ctx.stats()
    ms.append(st["kernel_ms"])
print("rows per band", rows, "kernel ms per band", np.round(ms, 3), "max/mean", round(max(ms) / np.mean(ms), 3), "sum", round(sum(ms), 2))
