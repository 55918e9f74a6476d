import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd as mi
S = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/scenes/cbox.xml")
res, spp = int(sys.argv[1]), int(sys.argv[2])
sc = mi.load_file(S, res=res, spp=spp)
integ = sc.integrator()
for _ in range(3): integ.render(sc, seed=0, spp=spp)
