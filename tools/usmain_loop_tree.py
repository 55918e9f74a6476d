#!/usr/bin/env python3
"""One finite-difference iteration of the reference's loop (USMain.py:279-283: 2 x (params.update + us_render)) timed on ANY tree
of this repository -- `python tools/usmain_loop_tree.py <tree root> [iterations]` -- so that an earlier round's library and Python
front end (git archive <commit> | tar -x -C _ab/rNN, make -C .../csrc) can be measured beside the current one in ONE GPU session.
Prints one JSON line: ms per iteration at paths_per_ray 1 / 64 / 4096."""
import importlib
import json
import os
import sys
import time

root = os.path.abspath(sys.argv[1])
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sys.path.insert(0, root)
mi = importlib.import_module("physics-based-ray-tracing_amd")
scene = mi.load_file(os.path.join(root, "tests", "scenes", "us_plate.xml"))
params = mi.traverse(scene)
key = [k for k in params.keys() if k.endswith("flat_plate.bsdf.roughness")][0]


def iteration(ppr, rough):
    for r in (rough, rough + 1e-3):
        params[key] = r
        params.update()
        img = mi.us_render(scene, seed=0, paths_per_ray=ppr)[0]
    return img


out = {"tree": root, "abi": mi._capi.PBRT_ABI_VERSION, "iterations": iters, "ms_per_iteration": {}}
for ppr in (1, 64, 4096):
    for _ in range(2):
        img = iteration(ppr, 0.1)
    t0 = time.perf_counter()
    for i in range(iters):
        iteration(ppr, 0.1 + 0.01 * i)
    out["ms_per_iteration"][str(ppr)] = round((time.perf_counter() - t0) / iters * 1e3, 3)
out["image"] = list(img.shape)
print(json.dumps(out))
