"""CPU restatement (numpy, f64) of the image-formation steps behind the ultrasound hot path -- TEST INFRASTRUCTURE
ONLY (imported by tests/, never by the product).  The reference delegates these steps to the third-party `ultraspy`
package at USMain.py:126-221; ultraspy (unpinned version) is absent from /root/reference and from this image, so the
arithmetic follows include/pbrt_hip.h's own definition of pbrt_das_beamform / pbrt_envelope / pbrt_log_compress.
PARITY UNPINNED: there is no reference output to pin these against; the tests pin them against closed forms
(point scatterer, pure tone, hand-computed decibels) and the HIP kernels against this file."""
from __future__ import annotations

import numpy as np


def das_beamform(data, tx_delays, elem_x, x, z, fs, sound_speed, t0=0.0, f_number=1.0, interpolation="linear", compound="sum"):
    """out[ix, iz] = sum_a sum_e data[a, e](t_tx(a; x, z) + |(x, z) - (elem_x[e], 0)| / c), t_tx = first arrival of the
    emitted wavefront (min over elements of delay + distance / c); receive aperture |x - x_e| <= z / (2 f_number)."""
    data = np.asarray(data, dtype=np.float32)
    A, E, T = data.shape
    tx = np.asarray(tx_delays, dtype=np.float32).reshape(A, E).astype(np.float64)
    ex = np.asarray(elem_x, dtype=np.float32).astype(np.float64)
    gx = np.asarray(x, dtype=np.float32).astype(np.float64)
    gz = np.asarray(z, dtype=np.float32).astype(np.float64)
    c = float(np.float32(sound_speed))
    fs = float(np.float32(fs))
    t0 = float(np.float32(t0))
    X, Z = np.meshgrid(gx, gz, indexing="ij")                       # [nx, nz]
    dist = np.sqrt((X[None] - ex[:, None, None]) ** 2 + Z[None] ** 2)  # [E, nx, nz]
    if f_number and f_number > 0:
        use = np.abs(X[None] - ex[:, None, None]) <= Z[None] / (2.0 * float(np.float32(f_number)))
    else:
        use = np.ones_like(dist, dtype=bool)
    out = np.zeros(X.shape, dtype=np.float64)
    for a in range(A):
        t_tx = np.min(tx[a][:, None, None] + dist / c, axis=0)     # [nx, nz]
        for e in range(E):
            s = (t_tx + dist[e] / c - t0) * fs
            tr = data[a, e].astype(np.float64)
            if interpolation == "nearest":
                r = np.rint(s)
                ok = (r >= 0) & (r <= T - 1) & use[e]
                out[ok] += tr[r[ok].astype(np.int64)]
            else:
                f = np.floor(s)
                ok = (f >= 0) & (f < T - 1) & use[e]
                i0 = f[ok].astype(np.int64)
                w = (s[ok] - f[ok]).astype(np.float32).astype(np.float64)
                out[ok] += tr[i0] + w * (tr[i0 + 1] - tr[i0])
                last = (s == T - 1) & use[e]
                out[last] += tr[T - 1]
    if compound == "mean":
        out /= A
    return out


def envelope(rf):
    """|analytic signal| along the last axis (the definition of scipy.signal.hilbert, written out)"""
    x = np.asarray(rf, dtype=np.float64)
    N = x.shape[-1]
    X = np.fft.fft(x, axis=-1)
    h = np.zeros(N)
    h[0] = 1.0
    if N % 2 == 0:
        h[N // 2] = 1.0
        h[1:N // 2] = 2.0
    else:
        h[1:(N + 1) // 2] = 2.0
    return np.abs(np.fft.ifft(X * h, axis=-1))


def log_compress(env, dynamic_range=60.0):
    """USMain.py:210-218"""
    e = np.asarray(env, dtype=np.float64)
    db = 20.0 * np.log10(e + 1e-12)
    mx = db.max()
    mn = mx - dynamic_range
    return (np.clip(db, mn, mx) - mn) / dynamic_range


def pulse_taps(fs, frequency, sigma):
    """h[k] = sin(2 pi fc k / fs) exp(-(k / fs)^2 / sigma^2), |k| <= ceil(2.5 sigma fs)   (RayTracingV0.py:194-198)"""
    fs, fc, sg = float(np.float32(fs)), float(np.float32(frequency)), float(np.float32(sigma))
    K = int(np.ceil(2.5 * sg * fs))
    t = np.arange(-K, K + 1) / fs
    return np.sin(2 * np.pi * fc * t) * np.exp(-(t * t) / (sg * sg)), K


def apply_pulse(traces, fs, frequency, sigma):
    """every trace (last axis) convolved with the pulse: out[n] = sum_k in[n - k] h[k], zero outside (pbrt_us_apply_pulse)"""
    x = np.asarray(traces, dtype=np.float64)
    h, K = pulse_taps(fs, frequency, sigma)
    flat = x.reshape(-1, x.shape[-1])
    out = np.stack([np.convolve(r, h, mode="full")[K:K + x.shape[-1]] for r in flat])
    return out.reshape(x.shape)
