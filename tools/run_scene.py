"""render one scene a few times and print the library's own statistics: python tools/run_scene.py scene.xml res spp [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd as mi
path, res, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
sc = mi.load_file(path, res=res, spp=spp)
for i in range(reps):
    mi.render(sc, seed=0)
    st = mi.default_context().stats()
print(f"{os.path.basename(path)} {res}x{res}x{spp}: kernel {st['kernel_ms']:.2f} ms = {res*res*spp/st['kernel_ms']/1e3:.0f} Msamples/s, "
      f"bounce {st['bounce_ms']:.2f} ms, segments/sample {st['segments']/st['samples']:.3f}, live {st['live'][:8]}, plan {st['fuse_plan']:#x}", flush=True)
