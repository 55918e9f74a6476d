"""ultrasound acquisition timing: python tools/us_scene.py scene.xml [paths_per_ray]   (US_SCENE_KW="k=v;k=v": load_file keywords)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd as mi
kw = dict(kv.split("=") for kv in os.environ.get("US_SCENE_KW", "").split(";") if kv)   # e.g. US_SCENE_KW="primary_rays=emitter"
us = mi.load_file(sys.argv[1], **kw)
ppr = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ui = us.integrator()
for i in range(3):
    ui._acquire(us, ui.quirks, paths_per_ray=ppr); st = mi.default_context().stats()
print(f"{os.path.basename(sys.argv[1])} {ui.n_angles}x{ui.n_elements}x{ppr}: kernel {st['kernel_ms']:.2f} ms = {st['samples']/st['kernel_ms']/1e3:.0f} Mpaths/s, "
      f"segments/path {st['segments']/st['samples']:.3f}, live {st['live'][:6]}", flush=True)
