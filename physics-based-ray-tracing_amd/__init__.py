"""MI355X-native Monte-Carlo ray-transport engine behind the Mitsuba-3 plugin API surface the
reference project (ReaganCardoza/Physics-Based-Ray-Tracing) is written against.

    import pbrt_amd as mi                      # instead of `import mitsuba as mi`
    mi.set_variant("hip_gfx950")               # USMain.py:12
    mi.register_bsdf("ultrasound_bsdf", mi.UltraBSDF)   # USMain.py:14-24 (already registered)
    scene = mi.load_dict(scene_dict)           # USMain.py:257
    scene.integrator().simulate_acquisition_parallel(scene)   # USMain.py:99
    img = mi.render(mi.load_file("scenes/cbox.xml", res=512, spp=256))

All transport runs in hand-written HIP kernels (csrc/, libpbrt_hip.so) reached through the C-ABI
of include/pbrt_hip.h via ctypes.  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np

from . import _capi, drjit_compat
from ._capi import (USQ_CLAMP_TIME, USQ_DIAG_SAMPLE, USQ_DOUBLE_LOCAL, USQ_MIXED_FRAMES, USQ_NEVER_ENTER,
                    USQ_NO_TOF_ACCUM, USQ_REF_REFLECT, USQ_REFERENCE, USQ_UNIT_GGX_PDF, Context, DeviceBuffer, HipLibraryMissing,
                    default_context, load_library)
from .plugins import (AreaEmitter, BSDF, BSDFContext, BSDFFlags, BSDFSample3f, ConductorBSDF, CustomEmitter,
                      DielectricBSDF, DiffuseBSDF, DirectIntegrator, DrArray, Emitter, EmitterFlags,
                      PathIntegrator, PerspectiveSensor, PointEmitter, SamplingIntegrator, Sensor,
                      SurfaceInteraction3f, UltraBSDF, UltraIntegrator, UltraSensor)
from .scene import (Film, Object, ParamFlags, ReconstructionFilter, Sampler, Scene, SceneParameters, Shape, load_dict,
                    load_file, register_bsdf, register_emitter, register_film, register_integrator, register_rfilter,
                    register_sampler, register_sensor, register_shape, traverse)
from .transforms import Properties, ScalarTransform4f, Transform4f
from .beamform import (DelayAndSum, GridScan, apply_pulse, build_probe, das_beamform, das_first_arrival, envelope, log_compress,
                       us_render)

# NB: the receive-side accumulator class `CustomSensor` is reached as pbrt_amd.CustomSensor.CustomSensor (module of
# the same name, like the reference's CustomSensor.py) or pbrt_amd.plugins.CustomSensor.
__version__ = "0.1.0"

_VARIANTS = ("hip_gfx950", "scalar_rgb", "scalar_mono", "llvm_ad_rgb", "llvm_ad_mono", "cuda_ad_rgb", "cuda_ad_mono")
_variant = "hip_gfx950"

Float = np.float32
UInt32 = np.uint32
Bool = np.bool_


def set_variant(*names):
    """mi.set_variant(...) (USMain.py:12, CustomEmmitter.py:4, TestScene.py:3).  Every Mitsuba variant
    name is accepted and maps to the one backend there is: HIP on gfx950."""
    global _variant
    for n in names:
        if n in _VARIANTS:
            _variant = "hip_gfx950"
            return
    raise ValueError(f"unknown variant {names}; known: {_VARIANTS}")


def variant():
    return _variant


def variants():
    return list(_VARIANTS)


def render(scene, params=None, sensor=0, integrator=None, seed=0, spp=0, **kw):
    """mi.render(scene, spp=..., seed=...) -> float32 [H, W, 3]"""
    integ = integrator or scene.integrator()
    if integ is None:
        raise ValueError("scene has no integrator")
    return integ.render(scene, sensor=sensor, seed=seed, spp=spp, **kw)


def Point3f(x, y=None, z=None):
    return np.asarray(x if y is None else [x, y, z], dtype=np.float32)


Vector3f = Point3f


class warp:
    """mi.warp.* used by the reference (CustomBSDF.py:48; UltraSensor): numpy
    restatements for host-side convenience (the kernels carry their own)."""

    @staticmethod
    def square_to_uniform_disk_concentric(sample):
        s = np.atleast_2d(np.asarray(sample, dtype=np.float64))
        x, y = 2 * s[:, 0] - 1, 2 * s[:, 1] - 1
        q = np.abs(x) < np.abs(y)
        r = np.where(q, y, x)
        rp = np.where(q, x, y)
        with np.errstate(divide="ignore", invalid="ignore"):
            phi = np.where((x == 0) & (y == 0), 0.0, 0.25 * np.pi * rp / r)
        phi = np.where(q, 0.5 * np.pi - phi, phi)
        return np.stack([r * np.cos(phi), r * np.sin(phi)], axis=1).astype(np.float32)

    @staticmethod
    def square_to_uniform_hemisphere(sample):
        p = warp.square_to_uniform_disk_concentric(sample).astype(np.float64)
        z = 1 - (p ** 2).sum(axis=1)
        p = p * np.sqrt(z + 1)[:, None]
        return np.concatenate([p, z[:, None]], axis=1).astype(np.float32)

    @staticmethod
    def square_to_cosine_hemisphere(sample):
        p = warp.square_to_uniform_disk_concentric(sample).astype(np.float64)
        z = np.sqrt(np.maximum(1 - (p ** 2).sum(axis=1), 0))
        return np.concatenate([p, z[:, None]], axis=1).astype(np.float32)


class Frame3f:
    """mi.Frame3f(n) (CustomBSDF.py:32-33): Duff et al. orthonormal basis, batched."""

    def __init__(self, n):
        n = np.atleast_2d(np.asarray(n, dtype=np.float64))
        sign = np.copysign(1.0, n[:, 2])
        a = -1.0 / (sign + n[:, 2])
        b = n[:, 0] * n[:, 1] * a
        self.s = np.stack([1 + sign * n[:, 0] ** 2 * a, sign * b, -sign * n[:, 0]], axis=1)
        self.t = np.stack([b, sign + n[:, 1] ** 2 * a, -n[:, 1]], axis=1)
        self.n = n

    def to_local(self, v):
        v = np.atleast_2d(np.asarray(v, dtype=np.float64))
        return np.stack([(v * self.s).sum(1), (v * self.t).sum(1), (v * self.n).sum(1)], axis=1).astype(np.float32)

    def to_world(self, v):
        v = np.atleast_2d(np.asarray(v, dtype=np.float64))
        return (self.s * v[:, 0:1] + self.t * v[:, 1:2] + self.n * v[:, 2:3]).astype(np.float32)
