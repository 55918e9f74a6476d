"""host-side overhead of one render call: wall time per call against the library's own GPU time (events), and where the Python part
goes.  usage: python tools/host_overhead.py [scene.xml res spp]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd as mi
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = sys.argv[1] if len(sys.argv) > 1 else "tests/scenes/cbox.xml"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sc = mi.load_file(os.path.join(root, scene), res=res, spp=spp)
integ = sc.integrator(); ctx = mi.default_context()
for _ in range(3): integ.render(sc, seed=0, spp=spp)
N = 30
t0 = time.perf_counter(); gpu = 0.0
for _ in range(N):
    integ.render(sc, seed=0, spp=spp); gpu += ctx.stats()["kernel_ms"]
wall = (time.perf_counter() - t0) * 1e3 / N
print(f"wall {wall:.3f} ms per call, GPU (ev0..ev1) {gpu / N:.3f} ms, host overhead {wall - gpu / N:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(N): integ.render(sc, seed=0, spp=spp)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
