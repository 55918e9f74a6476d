"""HIP image formation (pbrt_das_beamform / pbrt_envelope / pbrt_log_compress, through the C-ABI) against the numpy
restatement, the end-to-end B-mode of the reference's us_render, and its finite-difference roughness loop
(USMain.py:93-224, :257-289).  SURVEY section 8 f-1 / f-2."""
import numpy as np
import pytest

from conftest import scene_path
from oracle import beamform as obf

pytestmark = pytest.mark.gpu


def _rand_case(seed, A, E, T, c=1500.0, fs=20e6, pitch=3e-4):
    rng = np.random.default_rng(seed)
    data = rng.normal(size=(A, E, T)).astype(np.float32)
    ex = (pitch * (np.arange(E, dtype=np.float32) - (E - 1) / 2)).astype(np.float32)
    th = np.deg2rad(np.linspace(-12, 12, A))
    tx = (ex[None, :].astype(np.float64) * np.sin(th)[:, None] / c).astype(np.float32)
    return data, tx, ex, c, fs


@pytest.mark.parametrize("interp,fnum,compound", [("linear", 1.0, "sum"), ("linear", 0.0, "mean"), ("nearest", 1.5, "sum")])
def test_das_matches_the_restatement(mi, interp, fnum, compound):
    data, tx, ex, c, fs = _rand_case(1, 3, 16, 700)
    x = np.linspace(-0.004, 0.004, 37)
    z = np.linspace(0.0005, 0.024, 53)          # the deepest rows run off the end of the traces: zero contribution
    got = mi.das_beamform(data, tx, ex, x, z, fs, c, f_number=fnum, interpolation=interp, compound=compound)
    ref = obf.das_beamform(data, tx, ex, x, z, fs, c, f_number=fnum, interpolation=interp, compound=compound)
    assert got.shape == (37, 53) and got.dtype == np.float32
    if interp == "nearest":    # a sample position within 1e-9 of a half-integer may round either way: allow a handful
        bad = np.abs(got - ref) > 1e-4 * np.abs(ref).max()
        assert bad.mean() < 2e-3
    else:
        assert np.allclose(got, ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    assert np.abs(ref).max() > 1.0


def test_das_edges(mi):
    data, tx, ex, c, fs = _rand_case(2, 1, 4, 64)
    one = mi.das_beamform(data, tx, ex, [0.0], [0.001], fs, c, f_number=0.0)              # 1 x 1 grid
    assert one.shape == (1, 1) and one[0, 0] == pytest.approx(obf.das_beamform(data, tx, ex, [0.0], [0.001], fs, c, f_number=0.0)[0, 0], abs=1e-5)
    far = mi.das_beamform(data, tx, ex, [0.0, 0.001], [1.0, 2.0], fs, c)                  # beyond the recorded time: zeros
    assert np.array_equal(far, np.zeros((2, 2), np.float32))
    with pytest.raises(RuntimeError):
        mi.das_beamform(data, tx, ex, [0.0], [0.001], -1.0, c)                             # fs <= 0: error code, not a crash
    with pytest.raises(ValueError):
        mi.das_beamform(data[0], tx, ex, [0.0], [0.001], fs, c)


@pytest.mark.parametrize("N", [2, 64, 127, 398, 1024])
def test_envelope_matches_the_restatement(mi, N):
    rng = np.random.default_rng(N)
    rf = rng.normal(size=(7, N)).astype(np.float32)
    got = mi.envelope(rf)
    ref = obf.envelope(rf)
    assert got.shape == rf.shape and np.allclose(got, ref, rtol=0, atol=1e-4 * ref.max())
    n = np.arange(N)
    if N >= 64:
        tone = (0.5 * np.cos(2 * np.pi * 8 * n / N)).astype(np.float32)
        assert np.allclose(mi.envelope(tone[None])[0], 0.5, atol=2e-5)
    with pytest.raises(RuntimeError):
        mi.envelope(np.zeros((1, 5000), np.float32))                                       # nz > 4096


def test_log_compress_matches_the_restatement(mi):
    rng = np.random.default_rng(3)
    env = np.abs(rng.normal(size=(40, 50))).astype(np.float32) ** 4
    env[0, 0] = 0.0
    got = mi.log_compress(env, 60.0)
    assert got.shape == env.shape and np.allclose(got, obf.log_compress(env, 60.0), atol=2e-6)
    assert got.min() == 0.0 and got.max() == 1.0
    assert np.allclose(mi.log_compress(env, 40.0), obf.log_compress(env, 40.0), atol=2e-6)


def test_full_size_linearity(mi):
    """BASELINE-size input (5 x 64 x 10000 channel buffer, lambda / 4 grid of USMain.py:180-194): DAS is linear in the data"""
    data1, tx, ex, c, fs = _rand_case(5, 5, 64, 10000, c=1480.0, fs=50e6, pitch=1.2e-4)
    data2 = _rand_case(6, 5, 64, 10000)[0]
    lam = 1480.0 / 3e6
    x = np.arange(-0.04, 0.04 + lam / 4, lam / 4)
    z = np.arange(0.001, 0.05 + lam / 4, lam / 4)
    a = mi.das_beamform(data1, tx, ex, x, z, fs, c)
    b = mi.das_beamform(data2, tx, ex, x, z, fs, c)
    ab = mi.das_beamform(2.0 * data1 - 3.0 * data2, tx, ex, x, z, fs, c)
    assert a.shape == (len(x), len(z)) and np.abs(a).max() > 1
    assert np.allclose(ab, 2.0 * a - 3.0 * b, atol=2e-4 * np.abs(ab).max())
    env = mi.envelope(a)
    assert np.all(env >= np.abs(a) - 1e-3 * env.max())                                      # |analytic signal| >= |signal|
    band = slice(37, 41)                                                                    # a few columns against the restatement
    assert np.allclose(a[band], obf.das_beamform(data1, tx, ex, x[band], z, fs, c), atol=2e-5 * np.abs(a).max())


def test_bmode_of_the_plate_and_fd_roughness_loop(mi):
    """us_render end to end on the USMain.py scene, then the reference's finite-difference loop (:257-289) with
    common random numbers (same seed for every forward run -- the reference is unseeded and its FD gradient is noise)"""
    sc = mi.load_file(scene_path("us_plate.xml"))
    kw = dict(x_range=(-0.01, 0.01), z_range=(0.03, 0.07), seed=11, paths_per_ray=256)
    display, bmode, (xs, zs) = mi.us_render(sc, **kw)
    assert display.shape == (len(zs), len(xs)) and bmode.shape == (len(xs), len(zs))
    assert display.min() >= 0.0 and display.max() == 1.0 and np.isfinite(bmode).all()
    # the plate crosses the probe axis at z = 0.05 (45 degree tilt): the brightest pixel of the centre column is there
    col = bmode[len(xs) // 2]
    assert abs(zs[np.argmax(col)] - 0.05) < 0.004
    params = mi.traverse(sc)
    key = [k for k in params.keys() if k.endswith("flat_plate.bsdf.roughness")][0]

    def forward(rough):                                                                     # USMain.py:262-269
        params[key] = rough
        params.update()
        return mi.us_render(sc, **kw)[1]

    target = forward(0.7)
    again = forward(0.7)
    loss = lambda b: float(np.mean((b.astype(np.float64) - target) ** 2))                   # :273-274
    scale = float(np.mean(target.astype(np.float64) ** 2))
    assert loss(again) < 1e-8 * scale                                                       # reproducible up to the order of the atomics
    l3, l5 = loss(forward(0.3)), loss(forward(0.5))
    assert l3 > 1e-6 * scale and l5 > 1e-6 * scale and np.isfinite([l3, l5]).all()         # other roughness: another image
    rough, hist = 0.4, []
    for it in range(3):                                                                     # :279-289, step size normalised
        f0, f1 = loss(forward(rough)), loss(forward(rough + 1e-2))
        grad = (f1 - f0) / 1e-2
        g2 = (loss(forward(rough + 1e-2)) - loss(forward(rough))) / 1e-2                    # common random numbers: the
        assert np.isfinite(grad) and abs(grad - g2) <= 1e-5 * abs(grad) + 1e-12 * scale     # FD gradient is reproducible
        hist.append((rough, f0, grad))
        rough = float(np.clip(rough - 0.05 * np.sign(grad), 1e-4, 1.0))
    assert len(hist) == 3 and all(np.isfinite(h).all() for h in hist)


def test_apply_pulse_matches_the_restatement(mi):
    rng = np.random.default_rng(9)
    x = np.zeros((6, 3000), np.float32)
    idx = rng.integers(0, 3000, size=(6, 40))
    for r in range(6):
        x[r, idx[r]] = rng.normal(size=40).astype(np.float32)
    x[0, 0] = 1.0
    x[0, 2999] = -1.0                                                                       # pulses cut at both ends
    for fs, fc, sigma in ((50e6, 3e6, 5 / (4 * 3e6)), (50e6, 5e6, 2 / (4 * 5e6)), (20e6, 1e6, 3e-6)):
        got = mi.apply_pulse(x, fs, fc, sigma)
        ref = obf.apply_pulse(x, fs, fc, sigma)
        assert got.shape == x.shape and np.allclose(got, ref, rtol=0, atol=3e-5 * np.abs(ref).max())
    with pytest.raises(RuntimeError):
        mi.apply_pulse(x, 50e6, 3e6, 1e-3)                                                  # 125 000 taps: refused
    big = rng.normal(size=(5 * 64, 10000)).astype(np.float32)                               # the channel buffer of config 3
    a = mi.apply_pulse(big, 50e6, 3e6, 5 / (4 * 3e6))
    assert np.allclose(mi.apply_pulse(2 * big, 50e6, 3e6, 5 / (4 * 3e6)), 2 * a, atol=1e-4 * np.abs(a).max())
    assert np.allclose(a[17], obf.apply_pulse(big[17], 50e6, 3e6, 5 / (4 * 3e6)), atol=3e-5 * np.abs(a).max())


def test_gaussian_pulse_acquisition_matches_the_oracle(mi, ob):
    """pulse_model='gaussian': echoes deposited without the carrier (both sides), then the pulse on every trace"""
    sc = mi.load_file(scene_path("us_plate.xml"), paths_per_ray=64, seed=5)
    ui = sc.integrator()
    ref_impulse = ui._acquire(sc, ui.quirks)
    q = ui.quirks | mi._capi.USQ_NO_CARRIER
    got = ui._acquire(sc, q)
    osc = ob.OracleScene.from_scene(sc)
    amp, _ = osc.us_acquire(ui.us_params(sc, q), seed=5, paths_per_ray=64)
    ref = obf.apply_pulse(amp, ui.fs, ui.frequency, ui.pulse_sigma)
    assert got.shape == ref_impulse.shape == (5, 64, 10000)
    assert np.linalg.norm(got - ref) <= 1e-3 * np.linalg.norm(ref) and np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
    # each echo now covers ~ wave_cycles * fs / f samples instead of one
    assert (got != 0).sum() > 5 * (ref_impulse != 0).sum()


def test_device_resident_chain_equals_the_host_chain_bit_for_bit(mi):
    """ABI 5: acquisition -> (pulse) -> DAS -> envelope -> log compression queued on the context's stream on device pointers, one
    copy of the image to the host.  The channel buffer it ran on is then fetched and sent through the host-pointer entry points
    (round 4's chain: every step up and down PCIe): same kernels on the same data, the images are equal bit for bit.  Round 4's
    us_render itself (device_resident=False) runs its own acquisition, whose f32 atomics sum in another order: compared at the
    acquisition's tolerance."""
    for pulse in ("impulse", "gaussian"):
        sc = mi.load_file(scene_path("us_plate.xml"), paths_per_ray=32, seed=3)
        ui = sc.integrator()
        ui.pulse_model = pulse
        if pulse == "gaussian":
            ui.quirks |= mi._capi.USQ_NO_CARRIER
        kw = dict(x_range=(-0.012, 0.012), z_range=(0.03, 0.07))
        tm = {}
        d_dev, b_dev, (xs, zs) = mi.us_render(sc, timing=tm, **kw)
        assert ui._channel_host is None and ui._channel_dev is not None        # nothing but the two images crossed PCIe
        assert d_dev.shape == (len(zs), len(xs)) and b_dev.shape == (len(xs), len(zs))
        assert set(tm) == {"acquire", "queue", "wait_copy", "replayed"} and all(v >= 0 for v in tm.values()) and not tm["replayed"]
        chan = ui.channel_buf                                                   # the lazy copy happens here
        assert chan.shape == (5, 64, 10000) and ui._channel_host is not None and np.abs(chan).max() > 0
        probe = mi.build_probe("linear", 64, ui.pitch, ui.frequency, 70)
        rf = mi.das_beamform(chan, ui.transmission_delays_buf.reshape(5, 64), probe.geometry[0], xs, zs, ui.fs, ui.sound_speed)
        env = mi.envelope(rf)
        img = mi.log_compress(env, 60.0)
        assert np.array_equal(env, b_dev) and np.array_equal(img.T, d_dev)
        if pulse == "gaussian":                                                 # the buffer the images were formed from carries the pulse
            bare = ui._acquire(sc, ui.quirks, pulse=False)
            assert np.allclose(chan, mi.apply_pulse(bare, ui.fs, ui.frequency, ui.pulse_sigma), rtol=0, atol=2e-5 * np.abs(chan).max())
        d_host, b_host, _ = mi.us_render(sc, device_resident=False, **kw)
        assert np.allclose(b_host, b_dev, rtol=0, atol=2e-5 * b_dev.max())
        d_only, none, _ = mi.us_render(sc, return_bmode=False, **kw)
        assert none is None and d_only.shape == d_dev.shape and np.abs(d_only - d_dev).max() < 1e-3


def test_image_formation_on_device_buffers_step_by_step(mi):
    """each *_dev entry point against its host-pointer form on the same data (bit for bit), chained without a synchronisation in
    between; per-step device times come back when profiling is on"""
    cx = mi.default_context()
    data, tx, ex, c, fs = _rand_case(21, 5, 64, 4000, c=1540.0, fs=50e6, pitch=1.2e-4)
    lam = 1540.0 / 5e6
    x = np.arange(-0.01, 0.01 + lam / 4, lam / 4)
    z = np.arange(0.001, 0.03 + lam / 4, lam / 4)
    for interp, fnum, comp in (("linear", 1.0, "sum"), ("nearest", 0.0, "mean")):
        ref_bf = mi.das_beamform(data, tx, ex, x, z, fs, c, f_number=fnum, interpolation=interp, compound=comp)
        ref_env = mi.envelope(ref_bf)
        ref_img = mi.log_compress(ref_env, 50.0)
        cx.set_profiling(True)
        d_data = mi.DeviceBuffer.from_host(cx, data)
        d_bf = mi.das_beamform(d_data, tx, ex, x, z, fs, c, f_number=fnum, interpolation=interp, compound=comp)
        d_env = mi.envelope(d_bf)
        d_img = mi.log_compress(d_env, 50.0)
        assert isinstance(d_img, mi.DeviceBuffer) and d_img.shape == (len(x), len(z))
        img = d_img.numpy()                                                     # the first wait
        st = cx.image_stats()
        cx.set_profiling(False)
        assert np.array_equal(d_bf.numpy(), ref_bf) and np.array_equal(d_env.numpy(), ref_env) and np.array_equal(img, ref_img)
        assert st["measured"] & 0b1110 == 0b1110 and 0 < st["das_ms"] < 50 and 0 < st["envelope_ms"] < 50 and 0 < st["log_ms"] < 50
        assert st["das_model_bytes"] == (data.size + len(x) * len(z)) * 4
        # the first-arrival table of the scan, made once: the same image bit for bit without the pass over all elements
        tab = mi.das_first_arrival(tx, ex, x, z, c)
        assert tab.shape == (5, len(x), len(z)) and tab.dtype == np.float64
        d_bf2 = mi.das_beamform(d_data, tx, ex, x, z, fs, c, f_number=fnum, interpolation=interp, compound=comp, table=tab)
        assert np.array_equal(d_bf2.numpy(), ref_bf)
        t = tab.numpy()     # closed form of the plane-wave first arrival under the aperture: (x sin + z cos) / c
        th = np.deg2rad(np.linspace(-12, 12, 5))
        ix, iz = len(x) // 2, int(np.argmin(np.abs(z - 0.01)))   # (x - z tan(theta) inside the 7.7 mm array for every angle)
        assert np.allclose(t[:, ix, iz], (x[ix] * np.sin(th) + z[iz] * np.cos(th)) / c, rtol=0, atol=1e-9)
    pulsed = mi.apply_pulse(d_data, fs, 5e6, 5 / (4 * 5e6))
    assert np.array_equal(pulsed.numpy(), mi.apply_pulse(data, fs, 5e6, 5 / (4 * 5e6)))
    with pytest.raises(RuntimeError):
        mi.envelope(mi.DeviceBuffer(cx, (2, 5000)))                              # nz > 4096: error code from the *_dev form too
    with pytest.raises(RuntimeError):
        mi.apply_pulse(d_data, fs, 5e6, 5 / (4 * 5e6), out=d_data)               # in place is refused


def test_envelope_of_odd_and_even_columns_against_scipy_definition(mi):
    """the circular-convolution form of the analytic signal (round 5) at sizes around the tap-window edges: N = 1..9, 637 (the
    lambda / 4 grid of USMain.py at 5 MHz), 4095, 4096"""
    for N in list(range(1, 10)) + [637, 638, 4095, 4096]:
        rng = np.random.default_rng(100 + N)
        rf = rng.normal(size=(3, N)).astype(np.float32)
        ref = obf.envelope(rf)
        got = mi.envelope(rf)
        assert np.allclose(got, ref, rtol=0, atol=2e-4 * max(ref.max(), 1e-6)), N


def test_even_columns_skip_the_zero_taps_and_change_no_bit(mi, monkeypatch):
    """k_hilbert_env_even (even column lengths: every even tap of the discrete Hilbert kernel is zero, so an even output is a sum over
    the odd inputs only) against k_hilbert_env on the same columns: the same sums in the same order, bit for bit"""
    for N in (8, 10, 12, 14, 30, 256, 638, 1000, 4094, 4096):
        rng = np.random.default_rng(500 + N)
        rf = rng.normal(size=(5, N)).astype(np.float32)
        rf[1] = 0.0
        rf[2, N // 3] = 1e6                                 # (a spike: every output of the column sees one tap of it)
        got = mi.envelope(rf)
        monkeypatch.setenv("PBRT_ENV_GENERAL", "1")
        ref = mi.envelope(rf)
        monkeypatch.delenv("PBRT_ENV_GENERAL")
        assert np.array_equal(got, ref), N
        assert np.allclose(got, obf.envelope(rf), rtol=0, atol=2e-4 * np.abs(rf).max()), N


def test_a_queued_acquisition_reports_with_the_next_waiting_call(mi, capi):
    """pbrt_us_acquire_queue_dev (ABI 5): the call returns with the acquisition queued; the channel buffer is complete for the kernels
    queued behind it and for the host after a download; statistics (and a tripped guard) arrive with the next call that waits."""
    sc = mi.load_file(scene_path("us_plate.xml"), paths_per_ray=64, seed=9)
    ui = sc.integrator()
    cx = sc.device().ctx
    ref = ui._acquire(sc, ui.quirks)                                                      # host buffer, synchronous
    seg_ref = cx.stats()["segments"]
    d = mi.DeviceBuffer(cx, (ui.n_angles, ui.n_elements, ui.time_samples))
    ui._acquire(sc, ui.quirks, out_dev=d.ptr, queue=True)
    assert ui._ray_count is None                                                          # nothing has waited yet
    got = d.numpy()                                                                       # waits, finishes the queued call
    assert np.array_equal(got != 0, ref != 0) and np.allclose(got, ref, rtol=2e-5, atol=1e-7 * np.abs(ref).max())
    assert ui.ray_count == seg_ref == cx.stats()["segments"] and cx.stats()["samples"] == 5 * 64 * 64
    # two queued acquisitions back to back into two buffers, a material update between them (in stream order), one wait at the end
    d2 = mi.DeviceBuffer(cx, d.shape)
    params = mi.traverse(sc)
    key = [k for k in params.keys() if k.endswith("flat_plate.bsdf.roughness")][0]
    ui._acquire(sc, ui.quirks, out_dev=d.ptr, queue=True)
    params[key] = 0.3
    params.update()
    ui._acquire(sc, ui.quirks, out_dev=d2.ptr, queue=True)
    a, b = d.numpy(), d2.numpy()
    assert np.allclose(a, ref, rtol=2e-5, atol=1e-7 * np.abs(ref).max()) and not np.allclose(b, ref, rtol=1e-3, atol=1e-6 * np.abs(ref).max())
    params[key] = 0.7
    params.update()
    assert np.allclose(ui._acquire(sc, ui.quirks), ref, rtol=2e-5, atol=1e-7 * np.abs(ref).max())


def test_recorded_chain_replays_the_queued_calls(mi, capi):
    """pbrt_ctx_record_begin / _end / pbrt_graph_launch: the image-formation calls recorded on device buffers and replayed give
    the images of the plain calls bit for bit, also after the input buffer's CONTENTS changed; what may not run inside a recording
    says so and leaves it usable; a recording whose memory the context gave away refuses to launch."""
    cx = mi.default_context()
    data, tx, ex, c, fs = _rand_case(31, 5, 64, 4000, c=1540.0, fs=50e6, pitch=1.2e-4)
    lam = 1540.0 / 5e6
    x = np.arange(-0.01, 0.01 + lam / 4, lam / 4)
    z = np.arange(0.001, 0.03 + lam / 4, lam / 4)
    d_data = mi.DeviceBuffer.from_host(cx, data)
    tab = mi.das_first_arrival(tx, ex, x, z, c)
    d_tx, d_ex = mi.DeviceBuffer.from_host(cx, tx), mi.DeviceBuffer.from_host(cx, ex)
    d_x, d_z = mi.DeviceBuffer.from_host(cx, x.astype(np.float32)), mi.DeviceBuffer.from_host(cx, z.astype(np.float32))
    d_bf, d_env, d_img = (mi.DeviceBuffer(cx, (len(x), len(z))) for _ in range(3))

    def chain():
        mi.das_beamform(d_data, d_tx, d_ex, d_x, d_z, fs, c, out=d_bf, table=tab)
        mi.envelope(d_bf, out=d_env)
        mi.log_compress(d_env, 50.0, out=d_img)

    chain()                                                     # warm: the tap table and the block maxima exist now
    ref_env, ref_img = d_env.numpy(), d_img.numpy()
    with cx.record() as rec:
        chain()
        with pytest.raises(RuntimeError, match="recording"):
            cx.synchronize()                                    # nothing may wait inside a recording ...
        with pytest.raises(RuntimeError, match="recording"):
            d_img.numpy()
        with pytest.raises(RuntimeError, match="recording"):
            d_data.upload(data)
    g = rec.graph                                               # ... and the recording survived the refusals
    d_env.upload(np.zeros_like(ref_env))
    d_img.upload(np.zeros_like(ref_img))
    g.launch()
    assert np.array_equal(d_env.numpy(), ref_env) and np.array_equal(d_img.numpy(), ref_img)
    data2 = data[:, ::-1].copy()
    d_data.upload(data2)                                        # same pointers, other contents: the replay reads what is there now
    g.launch()
    got = d_img.numpy()
    d_data2 = mi.DeviceBuffer.from_host(cx, data2)
    mi.das_beamform(d_data2, d_tx, d_ex, d_x, d_z, fs, c, out=d_bf, table=tab)
    mi.envelope(d_bf, out=d_env)
    mi.log_compress(d_env, 50.0, out=d_img)
    assert np.array_equal(got, d_img.numpy()) and not np.array_equal(got, ref_img)
    # an envelope of another column length replaces the context's tap table: the recording is stale and says so
    mi.envelope(np.ones((2, 100), np.float32))
    with pytest.raises(RuntimeError, match="stale"):
        g.launch()
    g.close()
    # a recording that would have to allocate fails inside (a fresh context: no tap table yet), and the context works afterwards
    cx2 = capi.Context(cx.device)
    d_a, d_b = mi.DeviceBuffer.from_host(cx2, np.ones((3, 100), np.float32)), mi.DeviceBuffer(cx2, (3, 100))
    import ctypes as C
    h = capi._P()
    assert cx2.lib.pbrt_ctx_record_begin(cx2.handle) == 0
    assert cx2.lib.pbrt_envelope_dev(cx2.handle, 3, 100, capi._P(d_a.ptr), capi._P(d_b.ptr)) != 0
    assert cx2.lib.pbrt_ctx_record_end(cx2.handle, C.byref(h)) != 0 and not h.value     # reported again, no graph made
    assert b"inside the recording failed" in cx2.lib.pbrt_last_error(cx2.handle)
    with pytest.raises(RuntimeError, match="run the chain once"):
        with cx2.record():
            mi.envelope(d_a, out=d_b)
    mi.envelope(d_a, out=d_b)
    with cx2.record() as rec2:
        mi.envelope(d_a, out=d_b)
    d_b.upload(np.zeros((3, 100), np.float32))
    rec2.graph.launch()
    assert np.allclose(d_b.numpy(), mi.envelope(np.ones((3, 100), np.float32)), rtol=0, atol=1e-6)
    rec2.graph.close()
    for d in (d_a, d_b):
        d.close()
    cx2.close()
    chain()
    assert np.array_equal(d_img.numpy(), got)                 # (the context of the first recording still works)


def test_us_render_replays_its_chain_from_the_third_call_on(mi):
    """us_render on one scene with one set of arguments (the loop of USMain.py:262-289): call 1 queues the chain, call 2 queues and
    records it, calls 3.. replay the recording -- same image as the plain path at the acquisition's tolerance (f32 atomics), a
    material update between two replays is seen, other arguments fall back to the plain path, a context that profiles never
    replays."""
    sc = mi.load_file(scene_path("us_plate.xml"), paths_per_ray=16, seed=5)
    ui = sc.integrator()
    kw = dict(x_range=(-0.012, 0.012), z_range=(0.03, 0.07), return_bmode=True)
    imgs, flags = [], []
    for i in range(5):
        tm = {}
        d, b, _ = mi.us_render(sc, timing=tm, **kw)
        imgs.append(b)
        flags.append(tm["replayed"])
    assert flags == [False, False, True, True, True]
    st = sc.device().ctx.stats()                                               # of the last replay
    assert st["kernel_ms"] == 0.0 and st["samples"] == 5 * 64 * 16 and ui.ray_count == st["segments"] > 0   # (no event pairs in a replay)
    assert ui.transmission_delays_buf.shape == (5 * 64,)
    plain = mi.us_render(sc, graph=False, **kw)[1]
    assert sc.device().ctx.stats()["kernel_ms"] > 0.0
    for b in imgs:
        assert np.allclose(b, plain, rtol=0, atol=2e-5 * plain.max())
    # the finite-difference step: roughness changes in device memory, the replay sees it
    params = mi.traverse(sc)
    key = [k for k in params.keys() if k.endswith("flat_plate.bsdf.roughness")][0]
    params[key] = 0.3
    params.update()
    tm = {}
    rough = mi.us_render(sc, timing=tm, **kw)[1]
    assert tm["replayed"]
    rough_plain = mi.us_render(sc, graph=False, **kw)[1]
    assert np.allclose(rough, rough_plain, rtol=0, atol=2e-5 * rough_plain.max())
    assert not np.allclose(rough, plain, rtol=0, atol=1e-3 * plain.max())
    # another seed is another key: plain path, then its own recording
    tm = {}
    mi.us_render(sc, seed=77, timing=tm, **kw)
    assert not tm["replayed"]
    cx = sc.device().ctx
    cx.set_profiling(True)
    tm = {}
    mi.us_render(sc, timing=tm, **kw)
    das_ms = cx.image_stats()["das_ms"]
    cx.set_profiling(False)
    assert not tm["replayed"] and das_ms > 0


def test_us_render_hands_its_images_over_on_the_device(mi):
    """on_device=True: the display image and the envelope as DeviceBuffers, nothing copied, nothing waited for; torch reads them
    through __cuda_array_interface__ after a synchronise -- the same values as the host arrays of the plain call"""
    import torch
    sc = mi.load_file(scene_path("us_plate.xml"), paths_per_ray=8, seed=2)
    kw = dict(x_range=(-0.012, 0.012), z_range=(0.03, 0.07))
    disp, env, (xs, zs) = mi.us_render(sc, **kw)
    for _ in range(3):                                          # (the third call replays the recording)
        d_img, d_env, _ = mi.us_render(sc, on_device=True, **kw)
    assert isinstance(d_img, mi.DeviceBuffer) and d_img.shape == (len(xs), len(zs)) == d_env.shape
    sc.device().ctx.synchronize()
    t_img = torch.as_tensor(d_img, device="cuda")
    t_env = torch.as_tensor(d_env, device="cuda")
    assert t_img.data_ptr() == d_img.ptr and t_img.dtype == torch.float32 and tuple(t_img.shape) == d_img.shape
    assert np.allclose(t_img.T.cpu().numpy(), disp, rtol=0, atol=1e-3) and np.allclose(t_env.cpu().numpy(), env, rtol=0, atol=2e-5 * env.max())
    loss = torch.mean((t_img - torch.as_tensor(disp.T.copy(), device="cuda")) ** 2).item()   # a loss that never leaves the GPU
    assert 0.0 <= loss < 1e-6
