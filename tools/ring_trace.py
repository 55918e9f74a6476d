import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd as mi
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = mi.load_file(os.path.join(root, "tests/scenes/testring.xml"), res=1024, spp=64)
integ = sc.integrator()
for _ in range(2): integ.render(sc, seed=0, spp=64, flags=int(os.environ.get("PBRT_FLAGS", "0"), 0))
