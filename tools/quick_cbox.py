"""cbox 512^2 x 256 spp with the library's defaults: best-of-6 kernel ms (A/B of prebuilt libraries via PBRT_HIP_LIB)"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
sc = mi.load_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/scenes/cbox.xml"), res=512, spp=256)
integ = sc.integrator(); ctx = mi.default_context()
best = None
for _ in range(8):
    img = integ.render(sc, seed=0, spp=256, flags=int(os.environ.get("PBRT_FLAGS", "0"), 0))
    st = ctx.stats()
    if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
print(f"{os.path.basename(os.environ.get('PBRT_HIP_LIB', 'default'))} flags {os.environ.get('PBRT_FLAGS', '0')}: kernel {best['kernel_ms']:.3f} ms bounce {best['bounce_ms']:.3f} ms -> {512*512*256/best['kernel_ms']/1e3:.0f} Msamples/s  film sha {hashlib.sha1(img.tobytes()).hexdigest()[:10]} mean {img.mean():.6f}", flush=True)
