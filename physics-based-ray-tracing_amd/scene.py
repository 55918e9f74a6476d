"""Scene graph + plugin registry + parameter traversal: the host-side mirror of the Mitsuba-3
object model that the reference's driver uses (USMain.py:14-24 register_*, :257 load_dict,
:259-265 traverse / params.update()).  Everything here is bookkeeping; all ray transport goes to
libpbrt_hip.so through _capi (no CPU fallback)."""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _capi
from .meshio import load_mesh
from .transforms import Properties, ScalarTransform4f

# ------------------------------------------------------------------------------------------------
# plugin registry (mi.register_integrator / _bsdf / _emitter / _sensor, USMain.py:14-24)
# ------------------------------------------------------------------------------------------------
_REGISTRY: dict[str, tuple[str, type]] = {}


def _register(category, name, cls):
    _REGISTRY[name] = (category, cls)


def register_integrator(name, cls):
    _register("integrator", name, cls)


def register_bsdf(name, cls):
    _register("bsdf", name, cls)


def register_emitter(name, cls):
    _register("emitter", name, cls)


def register_sensor(name, cls):
    _register("sensor", name, cls)


def register_shape(name, cls):
    _register("shape", name, cls)


def register_film(name, cls):
    _register("film", name, cls)


def register_sampler(name, cls):
    _register("sampler", name, cls)


def register_rfilter(name, cls):
    _register("rfilter", name, cls)


class ParamFlags:
    NonDifferentiable = 0
    Differentiable = 1
    Discontinuous = 2


class Object:
    """Base of every plugin instance (mi.Object)."""
    category = "object"

    def __init__(self, props: Properties | None = None):
        self._id = props.id() if props is not None else ""

    def id(self):
        return self._id

    def traverse(self, callback):
        pass

    def parameters_changed(self, keys=None):
        pass

    def _children(self):
        """-> [(name, Object)] used by traverse()"""
        return []


def rgb3(v, default=None):
    if v is None:
        v = default
    if isinstance(v, dict):
        v = v.get("value", default)
    a = np.atleast_1d(np.asarray(v, dtype=np.float64)).ravel()
    if a.size == 1:
        a = np.repeat(a, 3)
    if a.size != 3:
        raise ValueError(f"expected an rgb triple, got {v!r}")
    return a


# ------------------------------------------------------------------------------------------------
# shapes -> pbrt_prim records
# ------------------------------------------------------------------------------------------------
class Shape(Object):
    category = "shape"

    def __init__(self, props):
        super().__init__(props)
        tw = props.get("to_world", None)
        self.to_world = tw if isinstance(tw, ScalarTransform4f) else ScalarTransform4f(tw) if tw is not None else ScalarTransform4f()
        self.flip_normals = bool(props.get("flip_normals", False))
        self._bsdf = None
        self._emitter = None
        for k in props.property_names():
            v = props.get(k)
            if isinstance(v, BSDFBase) and self._bsdf is None:
                self._bsdf = v
            elif isinstance(v, EmitterBase) and self._emitter is None:
                self._emitter = v
                v._shape = self

    def bsdf(self):
        return self._bsdf

    def emitter(self):
        return self._emitter

    def is_emitter(self):
        return self._emitter is not None

    def _children(self):
        out = []
        if self._bsdf is not None:
            out.append(("bsdf", self._bsdf))
        if self._emitter is not None:
            out.append(("emitter", self._emitter))
        return out

    def primitives(self) -> np.ndarray:
        raise NotImplementedError


def _tri_records(v0, v1, v2, ptype) -> np.ndarray:
    """float64 corner arrays [n,3] -> PRIM records with v0,e1,e2,n (n from the f32-rounded edges)"""
    n = len(v0)
    rec = np.zeros(n, dtype=_capi.PRIM_DTYPE)
    v0f = v0.astype(np.float32)
    e1 = (v1.astype(np.float32) - v0f).astype(np.float32)
    e2 = (v2.astype(np.float32) - v0f).astype(np.float32)
    nn = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    ln = np.linalg.norm(nn, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    nn = nn / ln
    rec["g"][:, 0:3] = v0f
    rec["g"][:, 3:6] = e1
    rec["g"][:, 6:9] = e2
    rec["g"][:, 9:12] = nn.astype(np.float32)
    rec["type"] = ptype
    return rec


class MeshShape(Shape):
    """'obj' / 'ply' shapes (scenes/cbox.xml:58-113, scenes/simple.xml:23-28)."""

    def __init__(self, props):
        super().__init__(props)
        self.filename = props["filename"]
        v, t, tn = load_mesh(self.filename, normals=True)
        self.vertices = self.to_world.transform_affine(v)
        self.faces = t
        # Mitsuba meshes: vertex normals, if the file has them and `face_normals` is not set, are interpolated into the
        # shading normal si.sh_frame.n = normalize(b0 n0 + b1 n1 + b2 n2); they go to world space with the inverse transpose
        self.face_normals = bool(props.get("face_normals", False))
        self.tri_normals = None
        if tn is not None and not self.face_normals:
            M = np.linalg.inv(self.to_world.matrix[:3, :3]).T
            w = tn @ M.T
            ln = np.linalg.norm(w, axis=2, keepdims=True)
            self.tri_normals = np.where(ln > 0, w / np.where(ln > 0, ln, 1.0), 0.0)        # [nt, 3, 3]
        self.merge_quads = bool(props.get("merge_quads", True))
        self.shading_normals = None   # [n_prims, 9] float32 after primitives(), or None

    def primitives(self):
        """Triangles; a fan pair (a,b,c),(a,c,d) whose f32 corners form an exact parallelogram
        (a + c == b + d component-wise) becomes ONE analytic PARALLELOGRAM primitive -- the same
        surface, one intersection test instead of two ("analytic quads", BASELINE config 2)."""
        v = self.vertices
        t = self.faces
        tn = self.tri_normals
        if self.flip_normals:
            t = t[:, [0, 2, 1]]
            if tn is not None:
                tn = -tn[:, [0, 2, 1]]
        vf = v.astype(np.float32)
        merged = np.zeros(len(t), bool)   # second triangle of a merged pair
        quad = np.zeros(len(t), bool)     # first triangle of a merged pair
        if self.merge_quads and len(t) >= 2:
            a, b, c = t[:-1, 0], t[:-1, 1], t[:-1, 2]
            a2, c2, d2 = t[1:, 0], t[1:, 1], t[1:, 2]
            cand = (a == a2) & (c == c2) & np.all(vf[a] + vf[c] == vf[b] + vf[d2], axis=1)
            if tn is not None:   # a quad keeps ONE shading normal: merge only where all its vertex normals agree
                tf = tn.astype(np.float32)
                same = np.all(tf[:-1] == tf[:-1, :1], axis=(1, 2)) & np.all(tf[1:] == tf[:-1, :1], axis=(1, 2))
                cand &= same
            k = 0
            while k < len(cand):      # greedy, non-overlapping pairs
                if cand[k]:
                    quad[k] = True
                    merged[k + 1] = True
                    k += 2
                else:
                    k += 1
        rec = _tri_records(v[t[:, 0]], v[t[:, 1]], v[t[:, 2]], _capi.PRIM_TRIANGLE)
        sn = tn.reshape(len(t), 9).astype(np.float32) if tn is not None else None
        if quad.any():
            qi = np.nonzero(quad)[0]
            q = _tri_records(v[t[qi, 0]], v[t[qi, 1]], v[t[qi + 1, 2]], _capi.PRIM_PARALLELOGRAM)
            rec[qi] = q
            rec = rec[~merged]
            if sn is not None:
                sn = sn[~merged]
        self.shading_normals = sn
        return rec


class SphereShape(Shape):
    """'sphere' (scenes/cbox.xml:115-129, MitsubaScenes/Sphere_Box.xml:36-45): centre/radius props
    and to_world (uniform scale)."""

    def __init__(self, props):
        super().__init__(props)
        c = np.asarray(props.get("center", [0.0, 0.0, 0.0]), dtype=np.float64)
        r = float(props.get("radius", 1.0))
        self.center = self.to_world.transform_affine(c)
        sx = np.linalg.norm(self.to_world.matrix[:3, 0])
        sy = np.linalg.norm(self.to_world.matrix[:3, 1])
        sz = np.linalg.norm(self.to_world.matrix[:3, 2])
        if not (abs(sx - sy) < 1e-6 * sx and abs(sx - sz) < 1e-6 * sx):
            raise ValueError("sphere: to_world must not contain non-uniform scale")
        self.radius = r * sx

    def primitives(self):
        rec = np.zeros(1, dtype=_capi.PRIM_DTYPE)
        rec["g"][0, 0:3] = self.center
        rec["g"][0, 3] = self.radius
        rec["type"] = _capi.PRIM_SPHERE
        return rec


class RectangleShape(Shape):
    """'rectangle' (USMain.py:67-90, MitsubaScenes/Sphere_Box.xml:47-101): the [-1,1]^2 square in
    the xy-plane, normal +z, under to_world -> one PARALLELOGRAM primitive."""

    def primitives(self):
        T = self.to_world
        c = T.transform_affine(np.array([[-1.0, -1.0, 0.0], [1.0, -1.0, 0.0], [-1.0, 1.0, 0.0]]))
        v0, v1, v2 = c[0:1], c[1:2], c[2:3]
        if self.flip_normals:
            v1, v2 = v2, v1
        return _tri_records(v0, v1, v2, _capi.PRIM_PARALLELOGRAM)


class ConeShape(Shape):
    """'cone' (MitsubaScenes/Cone_Box.xml:36-47, Cone_FLoating.xml) has no Mitsuba-3 definition.
    [DEFINE]: the closed unit cone -- apex (0, 0, 1), base disc of radius 1
    in the plane z = 0 -- under to_world (any affine map; Cone_Box.xml scales by 0.06 / 0.06 / 0.10).
    One analytic PRIM_CONE record carrying the world -> object matrix (the kernels intersect the quadric in object
    space), so a cone phantom stays on the brute-force path.  `tessellate=true` builds a triangle mesh instead
    (BVH path): `segments` around the axis, `rings` along it and across the base disc -- plain fans from the apex /
    the centre are 2 x `segments` slivers whose boxes all overlap, so both surfaces are cut into rings of quads with
    a fan only in the innermost ring.  (Measured on the Cone_Box phantom: 9.3 ms tessellated)"""

    def __init__(self, props):
        super().__init__(props)
        self.tessellate = bool(props.get("tessellate", False))
        self.segments = int(props.get("segments", 64))
        self.rings = int(props.get("rings", 4))
        if self.segments < 3 or self.rings < 1:
            raise ValueError("cone: segments must be >= 3 and rings >= 1")
        if abs(np.linalg.det(self.to_world.matrix[:3, :3])) < 1e-300:
            raise ValueError("cone: to_world must be invertible")

    def primitives(self):
        if not self.tessellate:
            rec = np.zeros(1, dtype=_capi.PRIM_DTYPE)
            M = self.to_world.matrix.astype(np.float64)
            if self.flip_normals:   # the record has no orientation bit; the mesh form has (triangle winding)
                raise NotImplementedError("cone: flip_normals needs tessellate=true")
            rec["g"][0] = np.linalg.inv(M)[:3, :4].reshape(12)
            rec["type"] = _capi.PRIM_CONE
            return rec
        return self._mesh_primitives()

    def _mesh_primitives(self):
        n, R = self.segments, self.rings
        ang = 2.0 * np.pi * np.arange(n, dtype=np.float64) / n
        c, s = np.cos(ang), np.sin(ang)

        def ring(radius, z):
            return np.stack([radius * c, radius * s, np.full(n, z)], axis=1)

        tris = []  # (v0, v1, v2) arrays [n, 3]
        apex = np.tile(np.array([[0.0, 0.0, 1.0]]), (n, 1))
        centre = np.zeros((n, 3))
        for k in range(R):       # lateral surface, level k -> k + 1 (z = k / R, radius 1 - z)
            lo = ring(1.0 - k / R, k / R)
            lo1 = np.roll(lo, -1, axis=0)
            if k == R - 1:
                tris.append((lo, lo1, apex))
            else:
                hi = ring(1.0 - (k + 1) / R, (k + 1) / R)
                hi1 = np.roll(hi, -1, axis=0)
                tris.append((lo, lo1, hi1))
                tris.append((lo, hi1, hi))
        for k in range(R):       # base disc (normal -z), radius k / R -> (k + 1) / R
            out = ring((k + 1) / R, 0.0)
            out1 = np.roll(out, -1, axis=0)
            if k == 0:
                tris.append((centre, out1, out))
            else:
                inn = ring(k / R, 0.0)
                inn1 = np.roll(inn, -1, axis=0)
                tris.append((inn, out1, out))
                tris.append((inn, inn1, out1))
        v0 = np.concatenate([t[0] for t in tris])
        v1 = np.concatenate([t[1] for t in tris])
        v2 = np.concatenate([t[2] for t in tris])
        T = self.to_world
        v0, v1, v2 = T.transform_affine(v0), T.transform_affine(v1), T.transform_affine(v2)
        if (np.linalg.det(T.matrix[:3, :3]) < 0) != self.flip_normals:
            v1, v2 = v2, v1
        return _tri_records(v0, v1, v2, _capi.PRIM_TRIANGLE)


# ------------------------------------------------------------------------------------------------
# base classes for BSDF / Emitter / Sensor / Integrator plugins
# ------------------------------------------------------------------------------------------------
class BSDFBase(Object):
    category = "bsdf"

    def to_material(self) -> tuple[int, list[float]]:
        raise NotImplementedError(
            f"{type(self).__name__} has no device material: the ray-transport hot path runs on the GPU only; "
            "a BSDF plugin must describe itself through to_material()")


class EmitterBase(Object):
    category = "emitter"
    _shape = None


class SensorBase(Object):
    category = "sensor"


class IntegratorBase(Object):
    category = "integrator"


class Film(Object):
    category = "film"

    def __init__(self, props):
        super().__init__(props)
        self.width = int(props.get("width", 768))
        self.height = int(props.get("height", 576))
        self.pixel_format = props.get("pixel_format", "rgb")
        self.component_format = props.get("component_format", "float32")
        self.rfilter = None
        for k in props.property_names():
            v = props.get(k)
            if isinstance(v, ReconstructionFilter):
                self.rfilter = v
        if self.rfilter is None:
            self.rfilter = ReconstructionFilter(Properties("gaussian"))  # hdrfilm default
        self.crop = (int(props.get("crop_offset_x", 0)), int(props.get("crop_offset_y", 0)),
                     int(props.get("crop_width", self.width)), int(props.get("crop_height", self.height)))

    def size(self):
        return (self.width, self.height)


class ReconstructionFilter(Object):
    category = "rfilter"
    _KINDS = {"box": _capi.FILTER_BOX, "tent": _capi.FILTER_TENT, "gaussian": _capi.FILTER_GAUSSIAN}

    def __init__(self, props):
        super().__init__(props)
        name = props.plugin_name()
        if name not in self._KINDS:
            raise NotImplementedError(f"reconstruction filter '{name}' is not supported")
        self.kind = self._KINDS[name]
        self.name = name


class Sampler(Object):
    category = "sampler"

    def __init__(self, props):
        super().__init__(props)
        self.sample_count = int(props.get("sample_count", 4))
        self.seed = int(props.get("seed", 0))
        # Integrator.sample(scene, sampler, ray): ray i draws from the key (index_offset + i, sample_index) under `seed`
        self.sample_index = int(props.get("sample_index", 0))
        self.index_offset = int(props.get("index_offset", 0))


# ------------------------------------------------------------------------------------------------
# Scene
# ------------------------------------------------------------------------------------------------
class Scene(Object):
    category = "scene"

    def __init__(self, objects: dict):
        super().__init__(None)
        self._objects = objects  # key -> Object, insertion order
        self._integrator = None
        self._sensors, self._shapes, self._emitters = [], [], []
        self._keys = {}
        for k, o in objects.items():
            self._keys[id(o)] = k
            if isinstance(o, IntegratorBase) and self._integrator is None:
                self._integrator = o
            elif isinstance(o, SensorBase):
                self._sensors.append(o)
            elif isinstance(o, Shape):
                self._shapes.append(o)
            elif isinstance(o, EmitterBase):
                self._emitters.append(o)
        self._flat = None
        self._dev = None
        self._dirty_materials = set()
        self.accel = _capi.ACCEL_AUTO

    # -- Mitsuba accessors used by the reference (CustomIntegrator.py:272, USMain.py:95)
    def integrator(self):
        return self._integrator

    def sensors(self):
        return self._sensors

    def shapes(self):
        return self._shapes

    def emitters(self):
        out = list(self._emitters)
        out += [s.emitter() for s in self._shapes if s.is_emitter()]
        return out

    def _children(self):
        return list(self._objects.items())

    # -- flattening to the C-ABI arrays -----------------------------------------------------------
    def flatten(self):
        """-> dict(prims, materials, emitters, light_prims, light_cdf, material_objects) of numpy
        arrays in the layout of include/pbrt_hip.h (pbrt_scene_desc)."""
        if self._flat is not None:
            return self._flat
        mats, mat_index = [], {}
        emitters = []
        prim_blocks, sn_blocks = [], []
        light_prims, light_cdf = [], []
        for si, sh in enumerate(self._shapes):
            rec = sh.primitives()
            sn = getattr(sh, "shading_normals", None)
            if sn is not None and sh.emitter() is not None:
                sn = None   # area lights shade and are sampled with their face normals (DESIGN D8)
            sn_blocks.append(sn if sn is not None else np.zeros((len(rec), 9), np.float32))
            b = sh.bsdf()
            if b is None:
                b = _default_bsdf()
            if id(b) not in mat_index:
                mat_index[id(b)] = len(mats)
                mats.append(b)
            rec["material"] = mat_index[id(b)]
            rec["shape"] = si
            rec["emitter"] = -1
            prim_blocks.append(rec)
        offsets = np.cumsum([0] + [len(r) for r in prim_blocks])
        for si, sh in enumerate(self._shapes):
            em = sh.emitter()
            if em is None:
                continue
            rec = prim_blocks[si]
            if np.any((rec["type"] == _capi.PRIM_SPHERE) | (rec["type"] == _capi.PRIM_CONE)):
                raise NotImplementedError("area emitters on spheres / analytic cones are not supported")
            e1 = rec["g"][:, 3:6].astype(np.float64)
            e2 = rec["g"][:, 6:9].astype(np.float64)
            area = np.linalg.norm(np.cross(e1, e2), axis=1)
            area = np.where(rec["type"] == _capi.PRIM_TRIANGLE, 0.5 * area, area)
            total = float(area.sum())
            cdf = np.cumsum(area) / total
            cdf[-1] = 1.0
            e = np.zeros(1, dtype=_capi.EMITTER_DTYPE)
            e["type"] = _capi.EMIT_AREA
            e["radiance"] = em.radiance_rgb().astype(np.float32)
            e["first"] = len(light_prims)
            e["count"] = len(rec)
            e["area"] = total
            rec["emitter"] = len(emitters)
            emitters.append(e)
            light_prims += list(range(offsets[si], offsets[si] + len(rec)))
            light_cdf += cdf.astype(np.float32).tolist()
        for em in self._emitters:
            if getattr(em, "is_transducer", False):  # CustomEmitter: a source of acoustic rays (UltraIntegrator primary_rays="emitter"), no light
                continue
            e = np.zeros(1, dtype=_capi.EMITTER_DTYPE)
            kind, rad, pos = em.device_emitter()
            e["type"] = kind
            e["radiance"] = np.asarray(rad, dtype=np.float32)
            e["pos"] = np.asarray(pos, dtype=np.float32)
            emitters.append(e)
        prims = np.concatenate(prim_blocks) if prim_blocks else np.zeros(0, dtype=_capi.PRIM_DTYPE)
        marr = np.zeros(len(mats), dtype=_capi.MATERIAL_DTYPE)
        for i, b in enumerate(mats):
            t, p = b.to_material()
            marr["type"][i] = t
            marr["p"][i, :len(p)] = p
        vnormals = np.ascontiguousarray(np.concatenate(sn_blocks), dtype=np.float32) if sn_blocks else np.zeros((0, 9), np.float32)
        self._flat = dict(
            prims=np.ascontiguousarray(prims), materials=marr, vertex_normals=vnormals if np.any(vnormals) else None,
            emitters=np.concatenate(emitters) if emitters else np.zeros(0, dtype=_capi.EMITTER_DTYPE),
            light_prims=np.asarray(light_prims, dtype=np.uint32), light_cdf=np.asarray(light_cdf, dtype=np.float32),
            material_objects=mats)
        return self._flat

    def device(self) -> "_capi.DeviceScene":
        """Upload (once) and return the device-resident scene.  Raises without the HIP library/GPU."""
        if self._dev is None:
            f = self.flatten()
            ctx = _capi.default_context()
            self._dev = _capi.DeviceScene(ctx, f["prims"], f["materials"], f["emitters"], f["light_prims"],
                                          f["light_cdf"], self.accel, f["vertex_normals"])
        if self._dirty_materials:
            f = self.flatten()
            for i in sorted(self._dirty_materials):
                t, p = f["material_objects"][i].to_material()
                f["materials"]["type"][i] = t
                f["materials"]["p"][i] = 0
                f["materials"]["p"][i, :len(p)] = p
                self._dev.update_material(i, _capi.make_material(t, p))
            self._dirty_materials.clear()
        return self._dev

    def _material_changed(self, bsdf):
        if self._flat is None:
            return
        for i, b in enumerate(self._flat["material_objects"]):
            if b is bsdf:
                self._dirty_materials.add(i)

    # -- batched scene queries (scene.ray_intersect, CustomIntegrator.py:309,324) ------------------
    def ray_intersect(self, o, d, tmax=None):
        """o, d: [n,3] world-space origins / unit directions.  -> dict(t, prim, u, v, valid, p, n, shape)."""
        o = np.atleast_2d(np.asarray(o, dtype=np.float32))
        d = np.atleast_2d(np.asarray(d, dtype=np.float32))
        n = len(o)
        tm = np.full(n, np.inf, dtype=np.float32) if tmax is None else _capi.f32(np.broadcast_to(tmax, (n,)))
        os_, ds_ = _capi.f32(o.T), _capi.f32(d.T)
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32)
        u = np.empty(n, np.float32)
        v = np.empty(n, np.float32)
        dev = self.device()
        dev.ctx.check(dev.ctx.lib.pbrt_ray_intersect(dev.handle, n, _capi.addr(os_), _capi.addr(ds_), _capi.addr(tm),
                                                      _capi.addr(t), _capi.addr(prim), _capi.addr(u), _capi.addr(v)),
                      "pbrt_ray_intersect")
        valid = prim != 0xFFFFFFFF
        out = dict(t=t, prim=prim, u=u, v=v, valid=valid)
        out.update(self._surface(o, d, t, prim, u, v, valid))
        return out

    def _surface(self, o, d, t, prim, u, v, valid):
        """hit point / normal / shape id from (prim,u,v,t): same formulas as the kernels."""
        P = self.flatten()["prims"]
        idx = np.where(valid, prim, 0).astype(np.int64)
        g = P["g"][idx]
        typ = P["type"][idx]
        p = g[:, 0:3] + u[:, None] * g[:, 3:6] + v[:, None] * g[:, 6:9]
        nrm = g[:, 9:12].copy()
        sph = (typ == _capi.PRIM_SPHERE) & valid   # misses carry t = inf
        if np.any(sph):
            ps = o + np.where(sph, t, 0.0)[:, None] * d
            ns = np.where(sph[:, None], ps - g[:, 0:3], 1.0)
            ns = ns / np.maximum(np.linalg.norm(ns, axis=1, keepdims=True), 1e-30)
            p = np.where(sph[:, None], g[:, 0:3] + ns * g[:, 3:4], p)
            nrm = np.where(sph[:, None], ns, nrm)
        cone = (typ == _capi.PRIM_CONE) & valid
        if np.any(cone):   # same formulas as make_si: p = o + t d, n = M^T (object-space gradient | -z)
            M = g.reshape(-1, 3, 4).astype(np.float64)
            pc = o + np.where(cone, t, 0.0)[:, None] * d
            q = np.einsum("nij,nj->ni", M[:, :, :3], pc) + M[:, :, 3]
            no = np.stack([q[:, 0], q[:, 1], 1.0 - q[:, 2]], axis=1)
            no[np.einsum("ni,ni->n", no, no) == 0] = [0.0, 0.0, 1.0]
            no[u != 0] = [0.0, 0.0, -1.0]
            nw = np.einsum("nji,nj->ni", M[:, :, :3], no)
            nw = nw / np.maximum(np.linalg.norm(nw, axis=1, keepdims=True), 1e-300)
            p = np.where(cone[:, None], pc, p)
            nrm = np.where(cone[:, None], nw, nrm)
        p[~valid] = 0
        nrm[~valid] = 0
        return dict(p=p.astype(np.float32), n=nrm.astype(np.float32), shape=np.where(valid, P["shape"][idx], -1))

    def ray_test(self, o, d, tmax=None):
        o = np.atleast_2d(np.asarray(o, dtype=np.float32))
        d = np.atleast_2d(np.asarray(d, dtype=np.float32))
        n = len(o)
        tm = np.full(n, np.inf, dtype=np.float32) if tmax is None else _capi.f32(np.broadcast_to(tmax, (n,)))
        os_, ds_ = _capi.f32(o.T), _capi.f32(d.T)
        hit = np.empty(n, np.uint8)
        dev = self.device()
        dev.ctx.check(dev.ctx.lib.pbrt_ray_test(dev.handle, n, _capi.addr(os_), _capi.addr(ds_), _capi.addr(tm),
                                                 _capi.addr(hit)), "pbrt_ray_test")
        return hit.astype(bool)

    def sample_emitter_direction(self, p, u):
        """Emitter.sample_direction for a batch: p [n,3] reference points, u [n,4] variates."""
        p = np.atleast_2d(np.asarray(p, dtype=np.float32))
        u = np.atleast_2d(np.asarray(u, dtype=np.float32))
        n = len(p)
        ps, us = _capi.f32(p.T), _capi.f32(u.T)
        d = np.empty((3, n), np.float32)
        q = np.empty((3, n), np.float32)
        w = np.empty((3, n), np.float32)
        dist = np.empty(n, np.float32)
        pdf = np.empty(n, np.float32)
        em = np.empty(n, np.uint32)
        dev = self.device()
        dev.ctx.check(dev.ctx.lib.pbrt_emitter_sample_direction(
            dev.handle, n, _capi.addr(ps), _capi.addr(us), _capi.addr(d), _capi.addr(dist), _capi.addr(pdf),
            _capi.addr(w), _capi.addr(q), _capi.addr(em)), "pbrt_emitter_sample_direction")
        return dict(d=d.T.copy(), dist=dist, pdf=pdf, weight=w.T.copy(), p=q.T.copy(), emitter=em)


def _default_bsdf():
    from .plugins import DiffuseBSDF
    return DiffuseBSDF(Properties("diffuse"))


# ------------------------------------------------------------------------------------------------
# load_dict / load_file (USMain.py:257; scenes/*.xml)
# ------------------------------------------------------------------------------------------------
def _instantiate(node, ids: dict, key: str = ""):
    if isinstance(node, Object):
        return node
    if not isinstance(node, dict) or "type" not in node:
        return node
    t = node["type"]
    if t == "ref":
        if node["id"] not in ids:
            raise KeyError(f'reference to unknown id "{node["id"]}"')
        return ids[node["id"]]
    if t in ("rgb", "spectrum"):
        return rgb3(node)
    if t == "scene":
        objs = {}
        for k, v in node.items():
            if k in ("type", "id"):
                continue
            o = _instantiate(v, ids, k)
            if isinstance(o, Object):
                objs[k] = o
        return Scene(objs)
    if t not in _REGISTRY:
        raise KeyError(f'unknown plugin type "{t}" (register it with register_integrator/_bsdf/_emitter/_sensor)')
    _, cls = _REGISTRY[t]
    props = Properties(t, {}, node.get("id", key))
    for k, v in node.items():
        if k in ("type", "id"):
            continue
        props[k] = _instantiate(v, ids, k)
    obj = cls(props)
    if getattr(obj, "_id", "") in ("", None):
        obj._id = props.id()
    if node.get("id"):
        ids[node["id"]] = obj
    elif key and key not in ids:
        ids[key] = obj
    return obj


def load_dict(d: dict):
    from . import plugins  # noqa: F401  (registers the built-ins)
    return _instantiate(d, {}, "")


def load_file(path: str, **kwargs):
    from .xml_loader import load_xml_to_dict
    return load_dict(load_xml_to_dict(path, **kwargs))


# ------------------------------------------------------------------------------------------------
# traverse (USMain.py:259-265)
# ------------------------------------------------------------------------------------------------
class SceneParameters:
    """params = traverse(scene); params['flat_plate.bsdf.roughness'] = 0.3; params.update()"""

    class _Collector:
        def __init__(self, owner, prefix, table):
            self.owner, self.prefix, self.table = owner, prefix, table

        def put_parameter(self, name, value, flags=ParamFlags.NonDifferentiable):
            self.table[self.prefix + name] = (self.owner, name, flags)

        # the reference's CustomSensor.traverse calls a method that does not exist in Mitsuba
        # (CustomSensor.py:68-73); accept it rather than crash
        put_parameters = put_parameter

        def put_object(self, name, obj, flags=ParamFlags.NonDifferentiable):
            pass

    def __init__(self, scene: Scene):
        self._scene = scene
        self._table = {}
        self._pending = {}
        self._walk(scene, "")

    def _walk(self, obj, prefix):
        for name, child in obj._children():
            cp = f"{prefix}{name}."
            child.traverse(self._Collector(child, cp, self._table))
            self._walk(child, cp)

    def _resolve(self, key):
        if key in self._table:
            return [key]
        # [DEFINE] the reference addresses 'shape.bsdf.roughness' (USMain.py:264) although its shapes are
        # keyed 'flat_plate' / 'wall_back': a leading 'shape.' addresses every shape of the scene.
        if key.startswith("shape."):
            rest = key[len("shape."):]
            hits = [k for k in self._table if k.split(".", 1)[-1] == rest and
                    isinstance(self._scene._objects.get(k.split(".", 1)[0]), Shape)]
            if hits:
                return hits
        raise KeyError(key)

    def keys(self):
        return list(self._table.keys())

    def __contains__(self, key):
        try:
            self._resolve(key)
            return True
        except KeyError:
            return False

    def __getitem__(self, key):
        k = self._resolve(key)[0]
        if k in self._pending:
            return self._pending[k]
        owner, name, _ = self._table[k]
        return getattr(owner, _attr_name(owner, name))

    def __setitem__(self, key, value):
        for k in self._resolve(key):
            self._pending[k] = value

    def update(self):
        changed = {}
        for k, v in self._pending.items():
            owner, name, _ = self._table[k]
            setattr(owner, _attr_name(owner, name), v)
            changed.setdefault(id(owner), (owner, []))[1].append(name)
        self._pending.clear()
        for owner, names in changed.values():
            owner.parameters_changed(names)
            if isinstance(owner, BSDFBase):
                self._scene._material_changed(owner)
        return list(changed)

    def __repr__(self):
        return "SceneParameters[\n  " + "\n  ".join(self._table.keys()) + "\n]"


def _attr_name(owner, name):
    return getattr(owner, "_param_attr", {}).get(name, name)


def traverse(scene: Scene) -> SceneParameters:
    return SceneParameters(scene)
