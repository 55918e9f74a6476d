"""k_das_beamform on sub-grids of the USMain.py scan (which part of the 150 us is what?): device time per call against the number of
8 x 8 tiles that see an element.  PBRT_HIP_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
rng = np.random.default_rng(0)
A, E, T, c, fs, pitch, fc = 5, 64, 10000, 1540.0, 50e6, 1.2e-4, 5e6
data = rng.normal(size=(A, E, T)).astype(np.float32)
ex = (pitch * (np.arange(E, dtype=np.float32) - (E - 1) / 2)).astype(np.float32)
tx = (ex[None, :].astype(np.float64) * np.sin(np.deg2rad([-15, -7.5, 0, 7.5, 15]))[:, None] / c).astype(np.float32)
lam = c / fc
cx = mi.default_context()
d_data, d_tx, d_ex = (mi.DeviceBuffer.from_host(cx, v) for v in (data, tx, ex))
cx.set_profiling(True)
for name, xr, zr, fnum in (("full scan", (-0.04, 0.04), (0.001, 0.05), 1.0), ("under the array", (-0.004, 0.004), (0.001, 0.05), 1.0),
                           ("deep half", (-0.04, 0.04), (0.025, 0.05), 1.0), ("shallow half", (-0.04, 0.04), (0.001, 0.025), 1.0),
                           ("outside the cone", (0.032, 0.04), (0.001, 0.02), 1.0), ("full scan, no aperture", (-0.04, 0.04), (0.001, 0.05), 0.0),
                           ("one tile row", (-0.04, 0.04), (0.04, 0.0405), 1.0)):
    x = np.arange(xr[0], xr[1] + lam / 4, lam / 4); z = np.arange(zr[0], zr[1] + lam / 4, lam / 4)
    X, Z = np.meshgrid(x, z, indexing="ij")
    half = Z / (2 * fnum) if fnum else np.full_like(Z, 1e9)
    inap = np.abs(X[None] - ex[:, None, None].astype(np.float64)) <= half[None]
    tiles = sum(inap[:, i:i + 8, k:k + 8].any() for i in range(0, len(x), 8) for k in range(0, len(z), 8))
    d_x, d_z = mi.DeviceBuffer.from_host(cx, x.astype(np.float32)), mi.DeviceBuffer.from_host(cx, z.astype(np.float32))
    out = mi.das_beamform(d_data, d_tx, d_ex, d_x, d_z, fs, c, f_number=fnum)
    t = 0.0
    for _ in range(5):
        mi.das_beamform(d_data, d_tx, d_ex, d_x, d_z, fs, c, f_number=fnum, out=out)
        t += cx.image_stats()["das_ms"] / 5
    print(f"{name:24s} {len(x):5d} x {len(z):4d} px  active tiles {tiles:5d} of {((len(x)+7)//8)*((len(z)+7)//8):5d}  pairs {int(inap.sum()):9d}  "
          f"das {t*1e3:7.1f} us  -> {t*1e6/max(tiles,1):6.3f} us per active tile-slot", flush=True)
