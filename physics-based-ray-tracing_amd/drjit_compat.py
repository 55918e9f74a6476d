"""The handful of `drjit` calls the reference's DRIVER code makes (USMain.py:41 dr.linspace,
CustomIntegrator.py:28,33 dr.arange/dr.linspace), returning numpy-backed arrays with .numpy().
    import pbrt_amd.drjit_compat as dr
"""
from __future__ import annotations

import numpy as np

pi = np.pi
inf = np.inf


def _dr(a):
    from .plugins import as_dr
    return as_dr(a)


def linspace(dtype, start, stop, num, endpoint=True):
    return _dr(np.linspace(start, stop, int(num), endpoint=endpoint, dtype=np.float32))


def arange(dtype, *args):
    return _dr(np.arange(*args, dtype=np.float32))


def zeros(dtype, shape=1):
    return _dr(np.zeros(shape, dtype=np.float32))


def full(dtype, value, shape=1):
    return _dr(np.full(shape, value, dtype=np.float32))


def deg2rad(x):
    return np.deg2rad(x)


def rad2deg(x):
    return np.rad2deg(x)


sin, cos, sqrt, abs, exp, floor = np.sin, np.cos, np.sqrt, np.abs, np.exp, np.floor
minimum, maximum = np.minimum, np.maximum


def clip(x, lo, hi):
    return np.clip(x, lo, hi)


clamp = clip
