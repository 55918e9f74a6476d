"""ad-hoc first GPU check: parity numbers + timing for the main paths"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrt_amd as mi
from oracle import binding as ob

def cmp(name, a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    print(f"{name}: shape={a.shape} rmse={np.sqrt(np.mean(d*d)):.3e} maxabs={np.abs(d).max():.3e} "
          f"exact={np.mean(a==b):.6f} ref_mean={b.mean():.4e} nan={np.isnan(a).sum()}", flush=True)

S = 'tests/scenes/'
for res, spp in ((64, 16), (128, 32)):
    sc = mi.load_file(S+'cbox.xml', res=res, spp=spp)
    t = time.time(); img = mi.render(sc, seed=3); t1 = time.time()-t
    integ, sens = sc.integrator(), sc.sensors()[0]
    ref = ob.OracleScene.from_scene(sc).render(sens.camera(), integ._film_desc(sc, sens, 3, spp), n_threads=16)
    cmp(f"cbox {res}x{spp}", img, ref); print(' t=', t1, mi.default_context().stats(), flush=True)
sc = mi.load_file(S+'simple.xml', res=64, spp=4)
img = mi.render(sc, seed=0); integ, sens = sc.integrator(), sc.sensors()[0]
ref = ob.OracleScene.from_scene(sc).render(sens.camera(), integ._film_desc(sc, sens, 0, 4), n_threads=16)
cmp("simple 64x4", img, ref); print(mi.default_context().stats(), flush=True)
sc = mi.load_file(S+'testring.xml', res=64, spp=8)
img = mi.render(sc, seed=0); integ, sens = sc.integrator(), sc.sensors()[0]
ref = ob.OracleScene.from_scene(sc).render(sens.camera(), integ._film_desc(sc, sens, 0, 8), n_threads=16)
cmp("testring 64x8", img, ref); print(mi.default_context().stats(), flush=True)
for name in ('us_plate.xml', 'us_sphere_box.xml'):
    us = mi.load_file(S+name, paths_per_ray=256)
    ui = us.integrator(); ui.simulate_acquisition_parallel(us)
    refb, _ = ob.OracleScene.from_scene(us).us_acquire(ui.us_params(us), ui.seed, 256)
    cmp(name, ui.channel_buf, refb); print(' nnz', (ui.channel_buf!=0).sum(), (refb!=0).sum(), mi.default_context().stats(), flush=True)
# perf: cbox 512 x 256
sc = mi.load_file(S+'cbox.xml', res=512, spp=256)
for i in range(3):
    t = time.time(); img = mi.render(sc, seed=0); dt = time.time()-t
    st = mi.default_context().stats()
    print(f"cbox 512x256: wall {dt*1e3:.1f} ms, kernel {st['kernel_ms']:.1f} ms, bounce {st['bounce_ms']:.1f} ms, Msamples/s {512*512*256/dt/1e6:.1f}", st, flush=True)
