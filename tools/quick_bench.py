"""quick A/B loop on the GPU box: golden check + cbox 512x512x256 timing (+ optional other configs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
S = 'tests/scenes/'
g = np.load('tests/golden/cbox_32x32_spp8_seed0.npy')
sc = mi.load_file(S + 'cbox.xml', res=32, spp=8)
img = mi.render(sc, seed=0)
print("golden cbox exact:", np.array_equal(img, g), "rmse", float(np.sqrt(np.mean((img - g) ** 2))), flush=True)
sc = mi.load_file(S + 'cbox.xml', res=512, spp=256)
best = 1e9
for i in range(int(os.environ.get("REPS", "5"))):
    t = time.time(); mi.render(sc, seed=0); dt = time.time() - t
    st = mi.default_context().stats()
    best = min(best, st['kernel_ms'])
print(f"cbox 512x256: best kernel {best:.2f} ms = {512*512*256/best/1e3:.0f} Msamples/s (bounce {st['bounce_ms']:.2f} ms, wall {dt*1e3:.1f})", flush=True)
if "ALL" in os.environ:
    sc = mi.load_file(S + 'testring.xml', res=1024, spp=64)
    for i in range(2):
        mi.render(sc, seed=0); st = mi.default_context().stats()
    print(f"testring 1024x64: kernel {st['kernel_ms']:.2f} ms = {1024*1024*64/st['kernel_ms']/1e3:.0f} Msamples/s", flush=True)
    us = mi.load_file(S + 'us_sphere_box.xml')
    ui = us.integrator()
    for i in range(2):
        ui._acquire(us, ui.quirks, paths_per_ray=65536); st = mi.default_context().stats()
    print(f"us_sphere_box 5x64x65536 paths: kernel {st['kernel_ms']:.2f} ms = {st['samples']/st['kernel_ms']/1e3:.0f} Mpaths/s segs/path {st['segments']/st['samples']:.2f}", flush=True)
    us = mi.load_file(S + 'us_cone_box.xml', tessellate="true")
    ui = us.integrator()
    for i in range(2):
        ui._acquire(us, ui.quirks, paths_per_ray=65536); st = mi.default_context().stats()
    print(f"us_cone_box (896 triangles) 5x64x65536 paths: kernel {st['kernel_ms']:.2f} ms = {st['samples']/st['kernel_ms']/1e3:.0f} Mpaths/s", flush=True)
