"""us_render eight times on a BVH phantom (k_trace / k_us_shade streams) and on the plate: which calls replay the recorded chain
(pbrt_graph_launch), wall-clock per call, same image every time (run on the GPU box)."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pbrt_amd as mi
for scene, ppr in (("tests/scenes/us_testring.xml", 16), ("tests/scenes/us_plate.xml", 1)):
    sc = mi.load_file(os.path.join(ROOT, scene), paths_per_ray=ppr, seed=1)
    flags, times = [], []
    ref = None
    for i in range(8):
        tm = {}
        t0 = time.perf_counter()
        d, b, _ = mi.us_render(sc, timing=tm, x_range=(-0.01, 0.01), z_range=(0.01, 0.04))
        times.append((time.perf_counter() - t0) * 1e3)
        flags.append(tm["replayed"])
        if ref is None: ref = b
        assert np.allclose(b, ref, rtol=0, atol=1e-4 * ref.max())
    print(scene, flags, [round(t, 3) for t in times], "ray_count", sc.integrator().ray_count, flush=True)
