"""kernel time of a render against the number of paths kept in flight per pass:
python tools/pass_size.py [scene.xml res spp mib,mib,...]   (default: the headline render, 2 .. 64 Mi)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
scene = sys.argv[1] if len(sys.argv) > 1 else "tests/scenes/cbox.xml"
res, spp = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 256)
sizes = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2, 4, 8, 16, 32, 64]
sc = mi.load_file(scene, res=res, spp=spp)
ref = None
for mib in sizes:
    best = 1e9
    for i in range(3):
        img = sc.integrator().render(sc, seed=0, spp=spp, pass_paths=mib << 20)
        st = mi.default_context().stats(); best = min(best, st["kernel_ms"])
    if ref is None: ref = img
    print(f"pass_paths {mib:4d} Mi: kernel {best:.2f} ms = {res * res * spp / best / 1e3:.0f} Msamples/s, bounce {st['bounce_ms']:.2f} ms, passes {st['passes']}, "
          f"launches {st['bounce_launches']}, workspace {st['workspace_bytes'] / 1e9:.1f} GB, same image {np.array_equal(img, ref)}", flush=True)
