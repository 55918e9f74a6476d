"""ctypes binding of libpbrt_hip.so (include/pbrt_hip.h).

This is the ONLY way the Python host reaches the HIP path.  There is no CPU fallback: if the
shared library is missing, or no MI355X is visible, every compute entry point raises.
The structures below mirror include/pbrt_hip.h field for field.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBRT_HIP_LIB") or os.path.join(_HERE, "csrc", "libpbrt_hip.so")  # override: A/B builds

PBRT_ABI_VERSION = 5

# primitive / material / emitter / filter / accel enums (include/pbrt_hip.h)
PRIM_TRIANGLE, PRIM_SPHERE, PRIM_PARALLELOGRAM, PRIM_CONE = 0, 1, 2, 3
MAT_DIFFUSE, MAT_CONDUCTOR, MAT_DIELECTRIC, MAT_ULTRA, MAT_NONE = 0, 1, 2, 3, 4
EMIT_AREA, EMIT_POINT = 0, 1
ACCEL_AUTO, ACCEL_BRUTE, ACCEL_BVH, ACCEL_BVH_GLOBAL = 0, 1, 2, 3
FILTER_BOX, FILTER_TENT, FILTER_GAUSSIAN = 0, 1, 2
FILM_RAW_ACCUM = 1
FILM_NO_REPACK = 2
FILM_NO_OCCLUDER_PRUNING = 4
FILM_NO_HIT_POOL = 0x10  # diagnostic: BVH scenes without k_bounce_pool
FILM_REGEN = 8  # diagnostic: brute-force scenes, persistent waves with path regeneration, one launch per pass (k_regen)
FILM_FUSE_PLAN_SET = 0x80


FILM_WALK_SET = 0x10000


def film_walk_from(depth: int) -> int:
    """flags value: from this depth on ONE launch walks every remaining bounce (0xff: never; include/pbrt_hip.h PBRT_FILM_WALK_FROM)"""
    return FILM_WALK_SET | ((int(depth) & 0xFF) << 17)


def film_fuse_plan(mask: int) -> int:
    """flags value: bit d of mask = the launch that walks bounce d goes on with bounce d + 1 in registers (include/pbrt_hip.h
    PBRT_FILM_FUSE_PLAN; 0 = one launch per bounce, 0x15 = pairs, 0xff = six bounces per launch)"""
    return FILM_FUSE_PLAN_SET | ((int(mask) & 0xFF) << 8)

US_MAX_ANGLES = 64

USQ_DIAG_SAMPLE = 0x1
USQ_REF_REFLECT = 0x2
USQ_UNIT_GGX_PDF = 0x4
USQ_DOUBLE_LOCAL = 0x8
USQ_MIXED_FRAMES = 0x10
USQ_NEVER_ENTER = 0x20
USQ_CLAMP_TIME = 0x40
USQ_NO_TOF_ACCUM = 0x80
USQ_NO_CARRIER = 0x100
USQ_NO_FIRST_TABLES = 0x200
USQ_NO_FUSED_BOUNCES = 0x400
USQ_FROZEN_DRAWS = 0x800
USQ_SIGNED_RR = 0x1000
USQ_DRJIT_VARIANT = USQ_CLAMP_TIME | USQ_NO_TOF_ACCUM | USQ_FROZEN_DRAWS | USQ_SIGNED_RR
USQ_REFERENCE = (USQ_DIAG_SAMPLE | USQ_REF_REFLECT | USQ_UNIT_GGX_PDF | USQ_DOUBLE_LOCAL
                 | USQ_MIXED_FRAMES | USQ_NEVER_ENTER)

PRIM_DTYPE = np.dtype([("g", np.float32, 12), ("type", np.uint32), ("material", np.uint32),
                       ("emitter", np.int32), ("shape", np.uint32)])
MATERIAL_DTYPE = np.dtype([("type", np.uint32), ("p", np.float32, 7)])
EMITTER_DTYPE = np.dtype([("type", np.uint32), ("radiance", np.float32, 3), ("pos", np.float32, 3),
                          ("first", np.uint32), ("count", np.uint32), ("area", np.float32),
                          ("pad", np.float32, 2)])
assert PRIM_DTYPE.itemsize == 64 and MATERIAL_DTYPE.itemsize == 32 and EMITTER_DTYPE.itemsize == 48


class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("p", C.c_float * 7)]


class SceneDesc(C.Structure):
    _fields_ = [("n_prims", C.c_uint32), ("prims", C.c_void_p),
                ("n_materials", C.c_uint32), ("materials", C.c_void_p),
                ("n_emitters", C.c_uint32), ("emitters", C.c_void_p),
                ("n_light_prims", C.c_uint32), ("light_prims", C.c_void_p), ("light_cdf", C.c_void_p),
                ("accel", C.c_uint32), ("vertex_normals", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("to_world", C.c_float * 12), ("tan_half_fov_x", C.c_float), ("near_clip", C.c_float),
                ("far_clip", C.c_float), ("film_w", C.c_uint32), ("film_h", C.c_uint32)]


class FilmDesc(C.Structure):
    _fields_ = [("crop_x", C.c_uint32), ("crop_y", C.c_uint32), ("crop_w", C.c_uint32), ("crop_h", C.c_uint32),
                ("spp", C.c_uint32), ("sample_offset", C.c_uint32), ("max_depth", C.c_uint32),
                ("rr_depth", C.c_uint32), ("filter", C.c_uint32), ("seed", C.c_uint32), ("flags", C.c_uint32),
                ("pass_paths", C.c_uint32)]


class UsEmitter(C.Structure):
    _fields_ = [("number_of_elements", C.c_uint32), ("pitch", C.c_float), ("element_width", C.c_float),
                ("element_height", C.c_float), ("radius", C.c_float), ("opening_angle", C.c_float),
                ("number_of_rays_per_element", C.c_uint32), ("speed_of_sound", C.c_float),
                ("steering_angle_min", C.c_float), ("steering_angle_max", C.c_float)]


class UsParams(C.Structure):
    _fields_ = [("max_depth", C.c_uint32), ("frequency", C.c_float), ("sound_speed", C.c_float),
                ("attenuation", C.c_float), ("main_beam_angle", C.c_float), ("cutoff_angle", C.c_float),
                ("fs", C.c_float), ("n_elements", C.c_uint32), ("pitch", C.c_float), ("n_angles", C.c_uint32),
                ("angles_deg", C.c_float * US_MAX_ANGLES), ("time_samples", C.c_uint32),
                ("sensor_to_world", C.c_float * 12), ("max_path_len", C.c_float), ("quirks", C.c_uint32),
                ("primary", C.c_uint32), ("emitter", UsEmitter)]


US_PRIMARY_ELEMENT, US_PRIMARY_EMITTER = 0, 1


class UsSensor(C.Structure):
    _fields_ = [("num_elements", C.c_uint32), ("element_width", C.c_float), ("element_height", C.c_float),
                ("pitch", C.c_float), ("radius", C.c_float), ("center_frequency", C.c_float),
                ("sound_speed", C.c_float), ("directivity", C.c_float), ("to_world", C.c_float * 12)]


class UsReceiver(C.Structure):
    _fields_ = [("number_of_elements", C.c_uint32), ("pitch", C.c_float), ("sample_rate", C.c_float),
                ("time_samples", C.c_uint32)]


class DasParams(C.Structure):
    _fields_ = [("n_angles", C.c_uint32), ("n_elements", C.c_uint32), ("time_samples", C.c_uint32), ("fs", C.c_float),
                ("sound_speed", C.c_float), ("t0", C.c_float), ("f_number", C.c_float), ("interpolation", C.c_uint32),
                ("compound_mean", C.c_uint32), ("nx", C.c_uint32), ("nz", C.c_uint32)]


DAS_NEAREST, DAS_LINEAR = 0, 1


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("kernel_ms", C.c_double), ("bounce_ms", C.c_double), ("bounce_launches", C.c_uint32),
                ("passes", C.c_uint32), ("model_bytes", C.c_uint64), ("bounce_model_bytes", C.c_uint64),
                ("live", C.c_uint64 * 16), ("fuse_plan", C.c_uint32), ("plan_source", C.c_uint32),
                ("pass_paths", C.c_uint64), ("workspace_bytes", C.c_uint64), ("trace_model_bytes", C.c_uint64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["live"] = list(self.live)
        return d


class ImageStats(C.Structure):
    _fields_ = [("pulse_ms", C.c_double), ("das_ms", C.c_double), ("envelope_ms", C.c_double), ("log_ms", C.c_double),
                ("das_model_bytes", C.c_uint64), ("measured", C.c_uint32), ("pad", C.c_uint32)]


_P = C.c_void_p
_F = C.c_void_p  # float* passed as raw address of a numpy buffer

# name -> (restype, argtypes); every symbol include/pbrt_hip.h declares
SIGNATURES = {
    "pbrt_abi_version": (C.c_int, []),
    "pbrt_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "pbrt_ctx_destroy": (C.c_int, [_P]),
    "pbrt_last_error": (C.c_char_p, [_P]),
    "pbrt_get_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "pbrt_ctx_set_workspace_limit": (C.c_int, [_P, C.c_uint64]),
    "pbrt_ctx_trim": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "pbrt_scene_create": (C.c_int, [_P, C.POINTER(SceneDesc), C.POINTER(_P)]),
    "pbrt_scene_update_material": (C.c_int, [_P, C.c_uint32, C.POINTER(Material)]),
    "pbrt_scene_destroy": (C.c_int, [_P]),
    "pbrt_render_radiance": (C.c_int, [_P, C.POINTER(Camera), C.POINTER(FilmDesc), _F]),
    "pbrt_render_radiance_dev": (C.c_int, [_P, C.POINTER(Camera), C.POINTER(FilmDesc), _P]),
    "pbrt_integrator_sample": (C.c_int, [_P, C.c_uint32, _F, _F, _F, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, _F]),
    "pbrt_us_acquire": (C.c_int, [_P, C.POINTER(UsParams), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _F, _F]),
    "pbrt_us_acquire_dev": (C.c_int, [_P, C.POINTER(UsParams), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, _F]),
    "pbrt_us_acquire_queue_dev": (C.c_int, [_P, C.POINTER(UsParams), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, _F]),
    "pbrt_ray_intersect": (C.c_int, [_P, C.c_uint32, _F, _F, _F, _F, _F, _F, _F]),
    "pbrt_ray_test": (C.c_int, [_P, C.c_uint32, _F, _F, _F, _F]),
    "pbrt_bsdf_sample": (C.c_int, [_P, C.POINTER(Material), C.c_uint32, C.c_uint32, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F]),
    "pbrt_bsdf_eval_pdf": (C.c_int, [_P, C.POINTER(Material), C.c_uint32, _F, _F, _F, _F]),
    "pbrt_emitter_sample_direction": (C.c_int, [_P, C.c_uint32, _F, _F, _F, _F, _F, _F, _F, _F]),
    "pbrt_sensor_sample_ray": (C.c_int, [_P, C.POINTER(Camera), C.c_uint32, _F, _F, _F, _F]),
    "pbrt_us_sensor_sample_ray": (C.c_int, [_P, C.POINTER(UsSensor), C.c_int, C.c_uint32, _F, _F, _F, _F, _F, _F, _F]),
    "pbrt_us_emitter_sample_ray": (C.c_int, [_P, C.POINTER(UsEmitter), C.c_uint32, _F, _F, _F, _F, _F, _F, _F, _F, _F]),
    "pbrt_us_put_data": (C.c_int, [_P, C.POINTER(UsReceiver), C.c_uint32, _F, _F, _F, _F, _F]),
    "pbrt_us_tx_delays": (C.c_int, [C.POINTER(UsParams), _F]),
    "pbrt_das_beamform": (C.c_int, [_P, C.POINTER(DasParams), _F, _F, _F, _F, _F, _F]),
    "pbrt_envelope": (C.c_int, [_P, C.c_uint32, C.c_uint32, _F, _F]),
    "pbrt_log_compress": (C.c_int, [_P, C.c_uint32, _F, C.c_float, _F]),
    "pbrt_us_apply_pulse": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, _F, _F]),
    # ABI 5: image formation on device pointers (queued on the ctx stream, no synchronisation), device buffers, the stream
    "pbrt_das_beamform_dev": (C.c_int, [_P, C.POINTER(DasParams), _P, _P, _P, _P, _P, _P]),
    "pbrt_das_first_arrival_dev": (C.c_int, [_P, C.POINTER(DasParams), _P, _P, _P, _P, _P]),
    "pbrt_das_beamform_table_dev": (C.c_int, [_P, C.POINTER(DasParams), _P, _P, _P, _P, _P, _P]),
    "pbrt_envelope_dev": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P]),
    "pbrt_log_compress_dev": (C.c_int, [_P, C.c_uint32, _P, C.c_float, _P]),
    "pbrt_us_apply_pulse_dev": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, _P, _P]),
    "pbrt_ctx_synchronize": (C.c_int, [_P]),
    "pbrt_dev_alloc": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "pbrt_dev_free": (C.c_int, [_P, _P]),
    "pbrt_dev_upload": (C.c_int, [_P, _P, _F, C.c_uint64]),
    "pbrt_dev_download": (C.c_int, [_P, _F, _P, C.c_uint64]),
    "pbrt_ctx_set_profiling": (C.c_int, [_P, C.c_int]),
    "pbrt_get_image_stats": (C.c_int, [_P, C.POINTER(ImageStats)]),
    "pbrt_ctx_record_begin": (C.c_int, [_P]),
    "pbrt_ctx_record_end": (C.c_int, [_P, C.POINTER(_P)]),
    "pbrt_graph_launch": (C.c_int, [_P]),
    "pbrt_graph_destroy": (C.c_int, [_P]),
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def _share_the_hip_runtime_with_torch():
    """One HIP runtime per process.  libpbrt_hip.so needs `libamdhip64.so.7`; a PyTorch-ROCm wheel bundles its own copy of
    that runtime (torch/lib/libamdhip64.so, same soname) and looks it up by FILE name, so if the system copy under
    /opt/rocm is mapped first a later `import torch` maps a second runtime, whose device enumeration then fails
    ("No HIP GPUs are available") -- and device pointers could not be exchanged between the two anyway (parallel.py
    renders into torch tensors).  If torch is installed, map ITS copy before ours: our NEEDED entry then resolves to it by
    soname, whichever of the two is imported first.  PBRT_HIP_RUNTIME=system keeps the system copy (no torch in the
    process)."""
    if os.environ.get("PBRT_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        rt = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(rt):
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # no torch, or not a ROCm build: the system runtime is the only one


def load_library(path: str | None = None):
    """dlopen libpbrt_hip.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise HipLibraryMissing(
            f"{p} not found: build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the ray-transport hot path.")
    _share_the_hip_runtime_with_torch()
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    ver = lib.pbrt_abi_version()
    if ver != PBRT_ABI_VERSION:
        raise RuntimeError(f"libpbrt_hip.so ABI {ver} != binding {PBRT_ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


@contextlib.contextmanager
def use_library(path: str):
    """Inside the block the package talks to ANOTHER build of the library (its own contexts; scenes loaded inside are bound to it):
    the diagnostic build libpbrt_hip_diag.so (make -C csrc diag: A/B launch structures the product library leaves out)."""
    global _lib, _default_ctx
    saved = (_lib, _default_ctx)
    _lib = load_library(path)
    _default_ctx = {}
    try:
        yield
    finally:
        _lib, _default_ctx = saved


DIAG_LIB_PATH = os.path.join(_HERE, "csrc", "libpbrt_hip_diag.so")
PLAN_CALLER, PLAN_LEARNT, PLAN_PROBED, PLAN_DEFAULT, PLAN_STREAMS = 0, 1, 2, 3, 4


def addr(a: np.ndarray | None):
    """Raw address of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def mat12(m) -> "C.Array":
    """3x4 row-major float[12] from a 4x4 (or 3x4) matrix."""
    a = np.asarray(m, dtype=np.float64)[:3, :4].astype(np.float32).ravel()
    return (C.c_float * 12)(*a.tolist())


def make_material(mtype: int, params) -> Material:
    m = Material()
    m.type = mtype
    p = list(params) + [0.0] * (7 - len(params))
    for i in range(7):
        m.p[i] = float(p[i])
    return m


class Context:
    """One pbrt_ctx per device (include/pbrt_hip.h).  Raises RuntimeError on any non-zero rc."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = _P()
        rc = self.lib.pbrt_ctx_create(int(device), C.byref(h))
        if rc != 0:
            msg = self.lib.pbrt_last_error(None)
            raise RuntimeError(f"pbrt_ctx_create(device={device}) failed rc={rc}: {msg.decode() if msg else ''}")
        self.handle = h
        self.device = device

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.pbrt_last_error(self.handle)
            raise RuntimeError(f"{what} failed rc={rc}: {msg.decode() if msg else ''}")

    def stats(self) -> dict:
        st = Stats()
        self.check(self.lib.pbrt_get_stats(self.handle, C.byref(st)), "pbrt_get_stats")
        return st.as_dict()

    def set_workspace_limit(self, n_bytes: int) -> None:
        """cap the device memory this context keeps for its calls (0: none); renders then take smaller passes"""
        self.check(self.lib.pbrt_ctx_set_workspace_limit(self.handle, C.c_uint64(int(n_bytes))), "pbrt_ctx_set_workspace_limit")

    def trim(self) -> int:
        """hand back what the last call did not need; -> bytes still held"""
        held = C.c_uint64(0)
        self.check(self.lib.pbrt_ctx_trim(self.handle, C.byref(held)), "pbrt_ctx_trim")
        return int(held.value)

    def synchronize(self) -> None:
        """wait for everything queued on the context's stream (the image-formation *_dev calls do not)"""
        self.check(self.lib.pbrt_ctx_synchronize(self.handle), "pbrt_ctx_synchronize")

    def set_profiling(self, on: bool) -> None:
        """HIP-event pairs around the image-formation steps (pbrt_get_image_stats); off by default"""
        self.check(self.lib.pbrt_ctx_set_profiling(self.handle, int(bool(on))), "pbrt_ctx_set_profiling")
        self.profiling = bool(on)

    def image_stats(self) -> dict:
        st = ImageStats()
        self.check(self.lib.pbrt_get_image_stats(self.handle, C.byref(st)), "pbrt_get_image_stats")
        return {k: getattr(st, k) for k, _ in st._fields_ if k != "pad"}

    def record(self) -> "Recording":
        """with ctx.record() as rec: <queueing calls> -- the calls are recorded instead of run (pbrt_ctx_record_begin / _end);
        rec.graph.launch() replays them in one submission"""
        return Recording(self)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pbrt_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Graph:
    """A recorded chain of queued calls (pbrt_graph): launch() submits all of it at once and returns without waiting."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self.handle = handle

    def launch(self) -> None:
        self.ctx.check(self.ctx.lib.pbrt_graph_launch(self.handle), "pbrt_graph_launch")

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.pbrt_graph_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Recording:
    """Context manager around pbrt_ctx_record_begin / pbrt_ctx_record_end; `graph` is set when the block ends without an error
    (an exception inside the block still closes the recording, and is re-raised)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.graph = None

    def __enter__(self):
        self.ctx.check(self.ctx.lib.pbrt_ctx_record_begin(self.ctx.handle), "pbrt_ctx_record_begin")
        return self

    def __exit__(self, et, ev, tb):
        h = _P()
        rc = self.ctx.lib.pbrt_ctx_record_end(self.ctx.handle, C.byref(h))
        if et is not None:
            if rc == 0:
                self.ctx.lib.pbrt_graph_destroy(h)
            return False
        self.ctx.check(rc, "pbrt_ctx_record_end")
        self.graph = Graph(self.ctx, h)
        return False


class DeviceBuffer:
    """A float32 / uint32 array in HBM owned by the library (pbrt_dev_alloc): what keeps the us_render loop on the device for a
    NumPy caller.  `ptr` is the device address the *_dev entry points take."""

    def __init__(self, ctx: Context, shape, dtype=np.float32):
        self.ctx = ctx
        self.shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        h = _P()
        ctx.check(ctx.lib.pbrt_dev_alloc(ctx.handle, C.c_uint64(self.nbytes), C.byref(h)), "pbrt_dev_alloc")
        self.ptr = h.value

    @classmethod
    def from_host(cls, ctx: Context, a) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(ctx, a.shape, a.dtype)
        b.upload(a)
        return b

    def upload(self, a) -> None:
        """in stream order; the source may be reused as soon as this returns"""
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.nbytes != self.nbytes:
            raise ValueError(f"upload of {a.nbytes} bytes into a buffer of {self.nbytes}")
        self.ctx.check(self.ctx.lib.pbrt_dev_upload(self.ctx.handle, _P(self.ptr), addr(a), C.c_uint64(self.nbytes)), "pbrt_dev_upload")

    @property
    def __cuda_array_interface__(self) -> dict:
        """zero-copy hand-over to a GPU array library of the same process (`torch.as_tensor(buf, device="cuda")`, CuPy): the
        consumer must order its work behind the context's stream itself (Context.synchronize(), or a stream wait)"""
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (int(self.ptr), False), "version": 2, "strides": None}

    def numpy(self) -> np.ndarray:
        """copy to the host (waits for the queued work)"""
        out = np.empty(self.shape, self.dtype)
        self.ctx.check(self.ctx.lib.pbrt_dev_download(self.ctx.handle, addr(out), _P(self.ptr), C.c_uint64(self.nbytes)), "pbrt_dev_download")
        return out

    def close(self):
        if getattr(self, "ptr", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.pbrt_dev_free(self.ctx.handle, _P(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: dict[int, Context] = {}


def default_context(device: int | None = None) -> Context:
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if "PBRT_DEVICE" not in os.environ else int(os.environ["PBRT_DEVICE"])
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def fill_scene_desc(prims: np.ndarray, materials: np.ndarray, emitters: np.ndarray, light_prims: np.ndarray,
                    light_cdf: np.ndarray, accel: int = ACCEL_AUTO, vertex_normals: np.ndarray | None = None) -> SceneDesc:
    """The arrays must stay alive while the returned descriptor is in use."""
    assert prims.dtype == PRIM_DTYPE and materials.dtype == MATERIAL_DTYPE and emitters.dtype == EMITTER_DTYPE
    d = SceneDesc()
    d.n_prims = len(prims)
    d.prims = addr(prims)
    d.n_materials = len(materials)
    d.materials = addr(materials)
    d.n_emitters = len(emitters)
    d.emitters = addr(emitters) if len(emitters) else None
    d.n_light_prims = len(light_prims)
    d.light_prims = addr(light_prims) if len(light_prims) else None
    d.light_cdf = addr(light_cdf) if len(light_cdf) else None
    d.accel = accel
    if vertex_normals is not None:
        assert vertex_normals.dtype == np.float32 and vertex_normals.shape == (len(prims), 9) and vertex_normals.flags["C_CONTIGUOUS"]
        d.vertex_normals = addr(vertex_normals)
    else:
        d.vertex_normals = None
    return d


class DeviceScene:
    """pbrt_scene handle: device-resident primitives / materials / emitters (+ BVH)."""

    def __init__(self, ctx: Context, prims, materials, emitters, light_prims, light_cdf, accel=ACCEL_AUTO, vertex_normals=None):
        self.ctx = ctx
        self._keep = (prims, materials, emitters, light_prims, light_cdf, vertex_normals)
        desc = fill_scene_desc(prims, materials, emitters, light_prims, light_cdf, accel, vertex_normals)
        h = _P()
        ctx.check(ctx.lib.pbrt_scene_create(ctx.handle, C.byref(desc), C.byref(h)), "pbrt_scene_create")
        self.handle = h
        self.n_prims = len(prims)

    def update_material(self, index: int, m: Material):
        self.ctx.check(self.ctx.lib.pbrt_scene_update_material(self.handle, int(index), C.byref(m)),
                       "pbrt_scene_update_material")

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx.lib.pbrt_scene_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
