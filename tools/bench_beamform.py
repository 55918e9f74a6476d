"""wall time of the image-formation steps at the reference's sizes (5 x 64 x 10000 channel buffer, lambda / 4 grid of
USMain.py:180-194: 651 x 399 pixels) through the host API (includes the 12.8 MB upload and the image download)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
rng = np.random.default_rng(0)
A, E, T, c, fs, pitch = 5, 64, 10000, 1480.0, 50e6, 1.2e-4
data = rng.normal(size=(A, E, T)).astype(np.float32)
ex = (pitch * (np.arange(E, dtype=np.float32) - (E - 1) / 2)).astype(np.float32)
tx = (ex[None, :].astype(np.float64) * np.sin(np.deg2rad([-15, -7.5, 0, 7.5, 15]))[:, None] / c).astype(np.float32)
lam = c / 3e6
x = np.arange(-0.04, 0.04 + lam / 4, lam / 4); z = np.arange(0.001, 0.05 + lam / 4, lam / 4)
for name, fn in (("das_beamform", lambda: mi.das_beamform(data, tx, ex, x, z, fs, c)),):
    img = fn()
    t = time.perf_counter(); [fn() for _ in range(5)]; dt = (time.perf_counter() - t) / 5
    print(f"{name}: {dt*1e3:.2f} ms per call, {len(x)} x {len(z)} pixels, {len(x)*len(z)*A*E/dt/1e9:.2f} G (pixel,angle,element) sums/s", flush=True)
env = mi.envelope(img)
t = time.perf_counter(); [mi.envelope(img) for _ in range(5)]; print(f"envelope: {(time.perf_counter()-t)/5*1e3:.2f} ms per call", flush=True)
t = time.perf_counter(); [mi.log_compress(env) for _ in range(5)]; print(f"log_compress: {(time.perf_counter()-t)/5*1e3:.2f} ms per call", flush=True)
