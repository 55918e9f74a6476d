"""Image formation behind the ultrasound hot path: delay-and-sum beamforming of the
channel buffer, envelope and log compression -- the second half of the reference's `us_render`
(USMain.py:93-224), which the reference delegates to the third-party `ultraspy` package.  `ultraspy` is not
available, so the arithmetic is this build's own definition (include/pbrt_hip.h, csrc/kernels_beamform.h,
restated in oracle/beamform.py; parity unpinned).  The classes keep the call shapes USMain.py uses:

    probe = build_probe(geometry_type='linear', nb_elements=N, pitch=p, central_freq=fc, bandwidth=70)   # :130-136
    beamformer = DelayAndSum(on_gpu=False); beamformer.automatic_setup(acquisition_info, probe)         # :174-175
    d_output = beamformer.beamform(d_data, GridScan(x_scan, z_scan))                                    # :204
    envelope = beamformer.compute_envelope(d_output, scan)                                              # :205

All work runs on the GPU through libpbrt_hip.so (no CPU fallback; `on_gpu` is accepted for compatibility)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi


def das_beamform(data, tx_delays, elem_x, x, z, fs, sound_speed, t0=0.0, f_number=1.0, interpolation="linear",
                 compound="sum") -> np.ndarray:
    """data [n_angles, n_elements, T] f32, tx_delays [n_angles, n_elements] (s), elem_x [n_elements] (m),
    grid x [nx], z [nz] (m)  ->  beamformed RF image [nx, nz] (pbrt_das_beamform)."""
    data = _capi.f32(np.asarray(data))
    if data.ndim != 3:
        raise ValueError("data must be [n_angles, n_elements, time_samples]")
    A, E, T = data.shape
    tx = _capi.f32(np.asarray(tx_delays).reshape(A, E))
    ex = _capi.f32(np.asarray(elem_x).reshape(E))
    gx, gz = _capi.f32(np.asarray(x).ravel()), _capi.f32(np.asarray(z).ravel())
    p = _capi.DasParams()
    p.n_angles, p.n_elements, p.time_samples = A, E, T
    p.fs, p.sound_speed, p.t0, p.f_number = float(fs), float(sound_speed), float(t0), float(f_number or 0.0)
    p.interpolation = {"nearest": _capi.DAS_NEAREST, "linear": _capi.DAS_LINEAR}[interpolation]
    p.compound_mean = {"sum": 0, "mean": 1}[compound]
    p.nx, p.nz = len(gx), len(gz)
    out = np.empty((p.nx, p.nz), dtype=np.float32)
    cx = _capi.default_context()
    cx.check(cx.lib.pbrt_das_beamform(cx.handle, C.byref(p), _capi.addr(data), _capi.addr(tx), _capi.addr(ex), _capi.addr(gx),
                                      _capi.addr(gz), _capi.addr(out)), "pbrt_das_beamform")
    return out


def envelope(rf) -> np.ndarray:
    """|analytic signal| along the last (axial) axis of a [nx, nz] RF image (pbrt_envelope)."""
    rf = _capi.f32(np.atleast_2d(np.asarray(rf)))
    nx, nz = rf.shape
    out = np.empty_like(rf)
    cx = _capi.default_context()
    cx.check(cx.lib.pbrt_envelope(cx.handle, nx, nz, _capi.addr(rf), _capi.addr(out)), "pbrt_envelope")
    return out


def log_compress(env, dynamic_range=60.0) -> np.ndarray:
    """USMain.py:210-218: 20 log10(env + 1e-12) clipped to the top `dynamic_range` dB, mapped to [0, 1]."""
    env = _capi.f32(np.asarray(env))
    out = np.empty_like(env)
    cx = _capi.default_context()
    cx.check(cx.lib.pbrt_log_compress(cx.handle, env.size, _capi.addr(env), float(dynamic_range), _capi.addr(out)),
             "pbrt_log_compress")
    return out


def apply_pulse(traces, fs, frequency, sigma) -> np.ndarray:
    """SURVEY f-3 pulse model (RayTracingV0.py:194-204): every trace (last axis) convolved with
    h[k] = sin(2 pi f k / fs) exp(-(k / fs)^2 / sigma^2)  (pbrt_us_apply_pulse)."""
    x = _capi.f32(np.asarray(traces))
    T = x.shape[-1]
    out = np.empty_like(x)
    cx = _capi.default_context()
    cx.check(cx.lib.pbrt_us_apply_pulse(cx.handle, x.size // T, T, float(fs), float(frequency), float(sigma), _capi.addr(x),
                                        _capi.addr(out)), "pbrt_us_apply_pulse")
    return out


# ---- ultraspy-shaped front end (USMain.py:126-205) ---------------------------------------------------------------
class Probe:
    def __init__(self, geometry_type, nb_elements, pitch, central_freq, bandwidth=70):
        if geometry_type != "linear":
            raise NotImplementedError("only the linear probe of USMain.py:130-136 is built (convex: SURVEY f-4)")
        self.geometry_type = geometry_type
        self.nb_elements = int(nb_elements)
        self.pitch = float(pitch)
        self.central_freq = float(central_freq)
        self.bandwidth = float(bandwidth)
        # same element positions as the integrator (CustomIntegrator.py:248)
        self.geometry = np.zeros((3, self.nb_elements), dtype=np.float32)
        self.geometry[0] = self.pitch * (np.arange(self.nb_elements, dtype=np.float32) - (self.nb_elements - 1) / 2)


def build_probe(geometry_type="linear", nb_elements=64, pitch=1.2e-4, central_freq=3e6, bandwidth=70):
    return Probe(geometry_type, nb_elements, pitch, central_freq, bandwidth)


class GridScan:
    def __init__(self, x_axis, z_axis):
        self.x_axis = np.asarray(x_axis, dtype=np.float64).ravel()
        self.z_axis = np.asarray(z_axis, dtype=np.float64).ravel()

    @property
    def shape(self):
        return (len(self.x_axis), len(self.z_axis))


class DelayAndSum:
    def __init__(self, on_gpu=True, f_number=1.0, interpolation="linear", compound="sum"):
        self.on_gpu = on_gpu  # accepted for compatibility; the beamformer has no CPU path
        self.setups = {"f_number": f_number, "interpolation": interpolation, "compound": compound}
        self.acquisition_info = None
        self.probe = None

    def automatic_setup(self, acquisition_info, probe):
        self.acquisition_info = dict(acquisition_info)
        self.probe = probe
        return self

    def update_setup(self, name, value):
        if name not in self.setups:
            raise KeyError(name)
        self.setups[name] = value

    def beamform(self, d_data, scan):
        ai = self.acquisition_info
        if ai is None or self.probe is None:
            raise RuntimeError("DelayAndSum.automatic_setup(acquisition_info, probe) has not been called")
        data = np.asarray(d_data)
        if data.ndim == 4:      # (frames, n_angles, n_elements, T): USMain passes reader.data[0]
            data = data[0]
        return das_beamform(data, ai["delays"], self.probe.geometry[0], scan.x_axis, scan.z_axis, ai["sampling_freq"],
                            ai["sound_speed"], t0=ai.get("t0", 0.0) or 0.0, f_number=self.setups["f_number"],
                            interpolation=self.setups["interpolation"], compound=self.setups["compound"])

    def compute_envelope(self, d_output, scan=None):
        return envelope(d_output)

    def __str__(self):
        return f"DelayAndSum(MI355X, {self.setups})"


def us_render(scene, x_range=(-0.04, 0.04), z_range=(0.001, 0.05), dynamic_range=60.0, step=None, seed=None,
              paths_per_ray=None, beamformer=None):
    """The reference's us_render (USMain.py:93-224) without the plotting: acquisition -> DAS -> envelope -> log
    compression.  Returns (display_image [nz, nx] in [0, 1], bmode envelope [nx, nz], (x_scan, z_scan))."""
    integ = scene.integrator()
    if seed is not None or paths_per_ray is not None:
        integ.channel_buf = integ._acquire(scene, integ.quirks, paths_per_ray=paths_per_ray, seed=seed)
    else:
        integ.simulate_acquisition_parallel(scene)                                                     # :99
    A, E, T = integ.n_angles, integ.n_elements, integ.time_samples
    data = np.asarray(integ.channel_buf, dtype=np.float32).reshape(A, E, T)                            # :118
    delays = np.asarray(integ.transmission_delays_buf, dtype=np.float32).reshape(A, E)                 # :121
    probe = build_probe("linear", E, integ.pitch, integ.frequency, 70)                                 # :130-136
    info = {"sampling_freq": integ.fs, "t0": 0, "prf": None, "signal_duration": None, "delays": delays,
            "sound_speed": integ.sound_speed,
            "sequence_elements": {"emitted": np.tile(np.arange(E), (A, 1)), "received": np.tile(np.arange(E), (A, 1))}}
    bf = beamformer or DelayAndSum(on_gpu=True)
    bf.automatic_setup(info, probe)                                                                    # :175
    lam = integ.sound_speed / integ.frequency
    step = step or lam / 4                                                                             # :189-191
    x_scan = np.arange(x_range[0], x_range[1] + step, step)                                            # :193
    z_scan = np.arange(z_range[0], z_range[1] + step, step)                                            # :194
    scan = GridScan(x_scan, z_scan)
    bmode = bf.compute_envelope(bf.beamform(data[np.newaxis], scan), scan).astype(np.float32)          # :204-207
    display = log_compress(bmode, dynamic_range).T                                                     # :210-221
    return display, bmode, (x_scan, z_scan)
