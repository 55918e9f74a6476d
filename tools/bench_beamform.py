"""Device time of the image-formation kernels at the sizes of the reference's us_render (USMain.py:26-90, :180-194: 5 x 64 x 10000
channel buffer, lambda / 4 grid at 5 MHz / 1540 m/s = 1040 x 638 pixels), data resident in HBM, HIP events on the library's
stream (pbrt_ctx_set_profiling).  PBRT_HIP_LIB selects the build; the md5 of each result says whether two builds agree."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
rng = np.random.default_rng(0)
A, E, T, c, fs, pitch, fc = 5, 64, 10000, 1540.0, 50e6, 1.2e-4, 5e6
data = rng.normal(size=(A, E, T)).astype(np.float32)
ex = (pitch * (np.arange(E, dtype=np.float32) - (E - 1) / 2)).astype(np.float32)
tx = (ex[None, :].astype(np.float64) * np.sin(np.deg2rad([-15, -7.5, 0, 7.5, 15]))[:, None] / c).astype(np.float32)
lam = c / fc
x = np.arange(-0.04, 0.04 + lam / 4, lam / 4); z = np.arange(0.001, 0.05 + lam / 4, lam / 4)
cx = mi.default_context()
d = {k: mi.DeviceBuffer.from_host(cx, v.astype(np.float32)) for k, v in dict(data=data, tx=tx, ex=ex, x=x, z=z).items()}
bf = mi.das_beamform(d["data"], d["tx"], d["ex"], d["x"], d["z"], fs, c)
env = mi.envelope(bf); img = mi.log_compress(env)
cx.set_profiling(True)
acc = dict(das_ms=0.0, envelope_ms=0.0, log_ms=0.0)
N = 10
for _ in range(N):
    mi.das_beamform(d["data"], d["tx"], d["ex"], d["x"], d["z"], fs, c, out=bf)
    mi.envelope(bf, out=env); mi.log_compress(env, out=img)
    st = cx.image_stats()
    for k in acc: acc[k] += st[k] / N
h = lambda b: hashlib.md5(b.numpy().tobytes()).hexdigest()[:10]
h_bf = h(bf)
# with the first-arrival table of the scan made once (pbrt_das_first_arrival_dev / pbrt_das_beamform_table_dev)
tab_ms = 0.0
if hasattr(mi, "das_first_arrival"):
    tab = mi.das_first_arrival(d["tx"], d["ex"], d["x"], d["z"], c)
    for _ in range(N):
        mi.das_beamform(d["data"], d["tx"], d["ex"], d["x"], d["z"], fs, c, out=bf, table=tab)
        tab_ms += cx.image_stats()["das_ms"] / N
print(f"{os.environ.get('PBRT_HIP_LIB', 'default'):40s} {len(x)} x {len(z)}  das {acc['das_ms']*1e3:7.1f} us (with table {tab_ms*1e3:6.1f} us, md5 {h(bf)})  envelope {acc['envelope_ms']*1e3:6.1f} us  "
      f"log {acc['log_ms']*1e3:5.1f} us   md5 das {h_bf} env {h(env)} img {h(img)}", flush=True)
