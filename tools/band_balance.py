"""Load balance of the band split of BASELINE config 5 (cbox 4096^2, 8 ranks) measured on ONE GPU: every virtual rank's
bands are rendered in turn (parallel.render_tiles), wall time and kernel time per rank; max / mean is what an 8-GPU strong
scaling run loses to imbalance before anything else.   usage: python tools/band_balance.py [spp] [band_rows ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pbrt_amd as mi
par = __import__("importlib").import_module("physics-based-ray-tracing_amd.parallel")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows_list = [int(x) for x in sys.argv[2:]] or [16, 32, 64, 128, 512]
RES, WORLD = 4096, 8
sc = mi.load_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/scenes/cbox.xml"), res=RES, spp=spp)
sc.device()
ctx = mi.default_context()
dev = torch.device("cuda", 0)
for rows in rows_list:
    wall, kern, calls = [], [], []
    for r in range(WORLD):
        acc = []
        par.render_tiles(sc, spp, 0, r, WORLD, rows, device=dev)            # warm (workspace sizes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        par.render_tiles(sc, spp, 0, r, WORLD, rows, device=dev, on_call=lambda: acc.append(ctx.stats()["kernel_ms"]))
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        kern.append(sum(acc))
        calls.append(len(acc))
    wall, kern = np.array(wall), np.array(kern)
    print(f"band_rows {rows:4d}: {calls[0]} calls per rank | wall ms per rank {np.round(wall, 1).tolist()} max/mean {wall.max() / wall.mean():.3f} | "
          f"kernel ms max/mean {kern.max() / kern.mean():.3f} | per-call overhead {(wall.sum() - kern.sum()) / sum(calls) * 1e3:.0f} us | "
          f"sum wall {wall.sum():.1f} ms -> {RES * RES * spp / wall.sum() / 1e3:.0f} Msamples/s on one GPU, ideal 8-GPU speed-up {wall.sum() / wall.max():.2f}x", flush=True)
