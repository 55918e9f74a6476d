"""Error behaviour of the C-ABI on a GPU box: invalid arguments come back as error codes (RuntimeError with the
library's message in the Python layer), never as crashes or silent CPU fallbacks; degenerate-but-valid inputs work."""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def test_invalid_render_arguments_are_refused(mi, capi):
    sc = mi.load_file(scene_path("cbox.xml"), res=16, spp=2)
    integ, sens = sc.integrator(), sc.sensors()[0]
    dev = sc.device()
    cam = sens.camera()
    out = np.empty((16, 16, 3), np.float32)

    def call(**kw):
        fd = integ._film_desc(sc, sens, 0, 2)
        for k, v in kw.items():
            setattr(fd, k, v)
        return dev.ctx.lib.pbrt_render_radiance(dev.handle, C.byref(cam), C.byref(fd), capi.addr(out))

    assert call() == 0
    for bad in (dict(crop_w=0), dict(crop_x=10, crop_w=10), dict(crop_y=16, crop_h=1), dict(spp=0), dict(max_depth=0), dict(filter=7)):
        rc = call(**bad)
        assert rc < 0, bad
        assert len(dev.ctx.lib.pbrt_last_error(dev.ctx.handle)) > 0
    assert dev.ctx.lib.pbrt_render_radiance(dev.handle, C.byref(cam), None, capi.addr(out)) < 0
    assert dev.ctx.lib.pbrt_render_radiance(None, C.byref(cam), None, capi.addr(out)) < 0
    assert call() == 0                                              # the context is still usable afterwards


def test_degenerate_but_valid_inputs(mi, ob):
    sc = mi.load_file(scene_path("cbox.xml"), res=16, spp=1)
    integ = sc.integrator()
    one = integ.render(sc, seed=3, spp=1, crop=(7, 9, 1, 1))         # 1 x 1 crop, 1 sample
    assert one.shape == (1, 1, 3) and np.isfinite(one).all()
    full = integ.render(sc, seed=3, spp=1)
    assert np.array_equal(one[0, 0], full[9, 7])
    # a scene whose camera sees nothing: every path misses at depth 0
    empty = mi.load_dict({"type": "scene", "integrator": {"type": "path", "max_depth": 3},
                          "sensor": {"type": "perspective", "to_world": mi.ScalarTransform4f().look_at([0, 0, 4], [0, 0, 5], [0, 1, 0]),
                                     "film": {"type": "hdrfilm", "width": 8, "height": 8, "rfilter": {"type": "box"}},
                                     "sampler": {"type": "independent", "sample_count": 2}},
                          "s": {"type": "sphere", "center": [0, 0, 0], "radius": 1.0, "bsdf": {"type": "diffuse"}},
                          "light": {"type": "point", "position": [0, 3, 0], "intensity": {"type": "rgb", "value": [1, 1, 1]}}})
    img = mi.render(empty, seed=0)
    assert img.shape == (8, 8, 3) and np.array_equal(img, np.zeros_like(img))
    st = mi.default_context().stats()
    assert st["segments"] == 0 and st["live"][0] == 128 and st["live"][1] == 0


def test_invalid_ultrasound_and_leaf_arguments(mi, capi):
    us = mi.load_file(scene_path("us_plate.xml"))
    ui = us.integrator()
    dev = us.device()
    p = ui.us_params(us)
    buf = np.empty((ui.n_angles, ui.n_elements, ui.time_samples), np.float32)
    tx = np.empty(ui.n_angles * ui.n_elements, np.float32)
    lib = dev.ctx.lib
    assert lib.pbrt_us_acquire(dev.handle, C.byref(p), 0, 4, 0, 4, capi.addr(buf), capi.addr(tx)) == 0
    p.n_angles = 0
    assert lib.pbrt_us_acquire(dev.handle, C.byref(p), 0, 4, 0, 4, capi.addr(buf), capi.addr(tx)) < 0
    p = ui.us_params(us)
    assert lib.pbrt_us_acquire(dev.handle, C.byref(p), 0, 0, 0, 4, capi.addr(buf), capi.addr(tx)) <= 0   # zero paths: nothing to do or refused
    assert lib.pbrt_us_acquire(dev.handle, None, 0, 4, 0, 4, capi.addr(buf), capi.addr(tx)) < 0
    o = np.zeros((3, 4), np.float32)
    assert lib.pbrt_ray_intersect(dev.handle, 4, capi.addr(o), None, None, None, None, None, None) < 0
    with pytest.raises(RuntimeError):
        mi.apply_pulse(np.zeros((2, 100), np.float32), 50e6, 3e6, -1.0)
    with pytest.raises(RuntimeError):
        mi.log_compress(np.ones(4, np.float32), dynamic_range=0.0)


def test_scenes_and_contexts_release_their_device_memory(mi):
    """create / render / destroy in a loop: the free device memory (hipMemGetInfo of the runtime the library itself is
    linked to) does not shrink -- scene buffers, BVH uploads and the context's work buffers are all released"""
    import ctypes
    import gc

    from conftest import scene_path

    mi.default_context()                       # maps the library and, with it, the one HIP runtime of this process
    path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64.so" in ln)
    hip = ctypes.CDLL(path)                    # the copy that is already mapped (torch's bundled one or the system's): the same instance

    def free_bytes():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    def cycle(n):
        for i in range(n):
            extra = mi.Context()               # a second context next to the default one: created and closed
            for scene, kw in (("cbox.xml", dict(res=32, spp=2)), ("testring.xml", dict(res=32, spp=1))):
                sc = mi.load_file(scene_path(scene), **kw)
                mi.render(sc, seed=i)          # uploads the scene (LDS-BVH image for the ring) and renders
                del sc                         # DeviceScene.__del__ -> pbrt_scene_destroy
            extra.close()
            gc.collect()

    cycle(3)                                   # warm-up: allocator pools, code objects
    free0 = free_bytes()
    cycle(25)
    free1 = free_bytes()
    assert free0 - free1 < 64 << 20            # 50 scenes with a 100 KB - 30 MB footprint each would show


def test_workspace_limit_and_trim(mi):
    """ABI 4: under a workspace limit a BVH scene renders the SAME film in more, smaller passes; a limit below the smallest pass
    is PBRT_E_NOMEM with a message (not a crash, not a silent truncation); pbrt_ctx_trim hands the memory back; without the limit
    the context grows again.  (The reference's driver imports torch beside the renderer, USMain.py:5: two allocators, one GPU.)"""
    ctx = mi.default_context()
    sc = mi.load_file(scene_path("testring.xml"), res=256, spp=64)
    try:
        ctx.set_workspace_limit(1)      # below whatever earlier tests left behind: everything goes back, the test starts from nothing
        assert ctx.trim() == 0
        ctx.set_workspace_limit(0)
        free = mi.render(sc, seed=3)
        st_free = ctx.stats()
        assert st_free["passes"] == 1 and st_free["workspace_bytes"] > 1_400_000_000      # 4 Mi paths x 340 B
        assert 1_400_000_000 < ctx.trim() <= st_free["workspace_bytes"]                    # the pass buffers fit the last call: they stay
        ctx.set_workspace_limit(600 << 20)                                                 # a limit below what is held: the 1.4 GB go back at once
        assert ctx.trim() == 0
        capped = mi.render(sc, seed=3)
        st = ctx.stats()
        assert st["passes"] == 4 and st["pass_paths"] == 256 * 256 * 16
        assert st["workspace_bytes"] <= 600 << 20
        assert np.array_equal(free, capped)
        # a brute-force scene and an acquisition under the same limit
        cb = mi.load_file(scene_path("cbox.xml"), res=64, spp=8)
        assert np.isfinite(mi.render(cb, seed=1)).all() and ctx.stats()["workspace_bytes"] <= 600 << 20
        # an acquisition on a mesh phantom (k_trace + k_us_shade share the radiance streams' workspace): three passes of 1 Mi paths
        # under the limit, one without it -- the same echoes (f32 sums in another order)
        us = mi.load_file(scene_path("us_testring.xml"))
        ui = us.integrator()
        few = ui._acquire(us, ui.quirks, paths_per_ray=8192, seed=2)
        assert ctx.stats()["passes"] >= 3 and ctx.stats()["workspace_bytes"] <= 600 << 20
        ctx.set_workspace_limit(0)
        one = ui._acquire(us, ui.quirks, paths_per_ray=8192, seed=2)
        assert ctx.stats()["passes"] == 1
        assert np.array_equal(few != 0, one != 0) and np.allclose(few, one, rtol=2e-5, atol=1e-7 * np.abs(one).max())
        ctx.set_workspace_limit(50 << 20)
        ctx.trim()
        with pytest.raises(RuntimeError, match="workspace limit"):
            mi.render(sc, seed=3)
        ctx.set_workspace_limit(0)
        again = mi.render(sc, seed=3)
        assert np.array_equal(free, again) and ctx.stats()["passes"] == 1
        held = ctx.stats()["workspace_bytes"]
        small = mi.load_file(scene_path("testring.xml"), res=32, spp=1)
        mi.render(small, seed=0)
        assert ctx.trim() < held // 8                                                      # the large pass buffers are not kept for a small call
    finally:
        ctx.set_workspace_limit(0)


def test_a_failed_allocation_halves_the_pass_of_an_acquisition(mi, monkeypatch):
    """ultrasound on a BVH scene sizes its pass from a free-memory snapshot; an allocation that fails afterwards
    (another allocator got there first -- PBRT_DEBUG_ALLOC_FAIL_BYTES makes every request above its size fail the way hipMalloc
    would) must halve the pass and go on like render_impl does, and when even the smallest pass does not fit, the call returns
    PBRT_E_NOMEM and hands back what it had already taken."""
    ctx = mi.default_context()
    us = mi.load_file(scene_path("us_testring.xml"))
    ui = us.integrator()
    try:
        ctx.set_workspace_limit(1)
        assert ctx.trim() == 0
        ctx.set_workspace_limit(0)
        one = ui._acquire(us, ui.quirks, paths_per_ray=16384, seed=4)               # 5 x 64 x 16384 = 5 Mi paths: one pass
        assert ctx.stats()["passes"] == 1
        ctx.set_workspace_limit(1)
        assert ctx.trim() == 0
        ctx.set_workspace_limit(0)
        monkeypatch.setenv("PBRT_DEBUG_ALLOC_FAIL_BYTES", str(100 << 20))           # a state plane set of 1 Mi paths (64 MB) fits, 2 Mi does not
        few = ui._acquire(us, ui.quirks, paths_per_ray=16384, seed=4)
        st = ctx.stats()
        assert st["passes"] >= 5 and st["workspace_bytes"] < 600 << 20
        assert np.array_equal(few != 0, one != 0) and np.allclose(few, one, rtol=2e-5, atol=1e-7 * np.abs(one).max())
        # the brute-force kernels' ping-pong state goes through the same loop
        sb = mi.load_file(scene_path("us_sphere_box.xml"))
        ub = sb.integrator()
        monkeypatch.delenv("PBRT_DEBUG_ALLOC_FAIL_BYTES")
        ref = ub._acquire(sb, ub.quirks, paths_per_ray=65536, seed=4)               # 20 Mi paths: two passes of 16 Mi
        monkeypatch.setenv("PBRT_DEBUG_ALLOC_FAIL_BYTES", str(300 << 20))           # 16 Mi x 60 B = 1 GB fails, 4 Mi paths fit
        ctx.set_workspace_limit(1)
        ctx.set_workspace_limit(0)
        got = ub._acquire(sb, ub.quirks, paths_per_ray=65536, seed=4)
        assert ctx.stats()["passes"] >= 5
        assert np.array_equal(got != 0, ref != 0) and np.allclose(got, ref, rtol=2e-5, atol=1e-7 * np.abs(ref).max())
        # nothing fits: an error code with a message, and the partial allocations are gone
        ctx.set_workspace_limit(1)
        ctx.set_workspace_limit(0)
        monkeypatch.setenv("PBRT_DEBUG_ALLOC_FAIL_BYTES", str(20 << 20))   # the channel buffer fits, no pass does
        with pytest.raises(RuntimeError, match="hipMalloc"):
            ui._acquire(us, ui.quirks, paths_per_ray=16384, seed=4)
        assert ctx.trim() < 64 << 20
    finally:
        monkeypatch.delenv("PBRT_DEBUG_ALLOC_FAIL_BYTES", raising=False)
        ctx.set_workspace_limit(0)


def test_acquisition_counters_survive_what_happens_between_two_acquisitions(mi):
    """the counter rows are zeroed by the reduction of the acquisition before (no fill command per call): a render, a trim that
    hands the rows back, a freed device buffer or another scene in between must not leave stale or uninitialised counters"""
    ctx = mi.default_context()
    us = mi.load_file(scene_path("us_plate.xml"), paths_per_ray=64, seed=4)
    ui = us.integrator()
    ui.simulate_acquisition_parallel(us)
    ref = ctx.stats()
    assert ref["samples"] == 5 * 64 * 64 and ref["segments"] > 0
    for between in ("nothing", "render", "trim", "dev_free", "other_acquisition"):
        if between == "render":
            mi.render(mi.load_file(scene_path("cbox.xml"), res=32, spp=4), seed=1)
        elif between == "trim":
            mi.render(mi.load_file(scene_path("cbox.xml"), res=32, spp=4), seed=1)
            ctx.trim()                                        # the acquisition's buffers were not used by the last call: they go
        elif between == "dev_free":
            mi.DeviceBuffer(ctx, (1024,)).close()
        elif between == "other_acquisition":
            other = mi.load_file(scene_path("us_sphere_box.xml"), paths_per_ray=256, seed=1)
            other.integrator().simulate_acquisition_parallel(other)
            assert ctx.stats()["samples"] == 5 * 64 * 256
        ui.simulate_acquisition_parallel(us)
        st = ctx.stats()
        assert (st["samples"], st["segments"], st["shadow_rays"], st["live"]) == (ref["samples"], ref["segments"], ref["shadow_rays"], ref["live"]), between
