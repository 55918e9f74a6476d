"""Plugin classes: the four plugin types of the reference (UltraIntegrator, UltraBSDF, CustomEmitter,
CustomSensor / UltraSensor) with the reference's constructor properties, attributes and method
names, plus the Mitsuba built-ins its scene files name (path, direct, diffuse, conductor,
dielectric, area, point, perspective, hdrfilm, independent, box/tent/gaussian, obj/ply/sphere/
rectangle).  Every method that computes is a batched array call into libpbrt_hip.so; nothing here
evaluates transport on the CPU."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _capi
from .scene import (BSDFBase, ConeShape, EmitterBase, Film, IntegratorBase, MeshShape, ParamFlags,
                    ReconstructionFilter, RectangleShape, Sampler, SensorBase, SphereShape, rgb3, _register)
from .transforms import Properties, ScalarTransform4f


class DrArray(np.ndarray):
    """ndarray with the .numpy() accessor the reference's driver calls (USMain.py:112)."""

    def numpy(self):
        return np.asarray(self)


def as_dr(a, dtype=np.float32):
    return np.asarray(a, dtype=dtype).view(DrArray)


def _soa3(a, n=None):
    a = np.atleast_2d(np.asarray(a, dtype=np.float32))
    if n is not None and len(a) == 1 and n > 1:
        a = np.broadcast_to(a, (n, 3))
    return _capi.f32(a.T)


def _vec(a, n):
    return _capi.f32(np.broadcast_to(np.asarray(a, dtype=np.float32), (n,)))


# ================================================================================================
# BSDFs
# ================================================================================================
class BSDFFlags:
    Null = 0x1
    DiffuseReflection = 0x2
    GlossyReflection = 0x8
    DeltaReflection = 0x20
    GlossyTransmission = 0x10
    DeltaTransmission = 0x40
    FrontSide = 0x10000
    BackSide = 0x20000
    Smooth = 0x2 | 0x8 | 0x10
    Delta = 0x1 | 0x20 | 0x40


class BSDFContext:
    def __init__(self, mode="radiance"):
        self.mode = mode


class BSDFSample3f:
    """Result record of BSDF.sample (CustomBSDF.py:160-168), arrays of length n."""

    def __init__(self, wo, pdf, eta, sampled_type, sampled_component):
        self.wo, self.pdf, self.eta = wo, pdf, eta
        self.sampled_type, self.sampled_component = sampled_type, sampled_component


class SurfaceInteraction3f:
    """Batched surface interaction: the fields BSDF.sample reads (CustomBSDF.py:90-95): wi (local
    shading frame), n (world geometric normal), sh_frame.n (world shading normal)."""

    class _Frame:
        def __init__(self, n):
            self.n = n

    def __init__(self, wi, n=None, sh_n=None, sh_s=None):
        """sh_s: tangent of the shading frame (sh_frame.s, or the shape's dp_du: Mitsuba orthonormalises it against
        sh_frame.n); None = coordinate_system(sh_frame.n)"""
        self.wi = np.atleast_2d(np.asarray(wi, dtype=np.float32))
        cnt = len(self.wi)
        z = np.tile(np.array([[0, 0, 1]], np.float32), (cnt, 1))
        self.n = np.atleast_2d(np.asarray(n, dtype=np.float32)) if n is not None else z
        self.sh_frame = self._Frame(np.atleast_2d(np.asarray(sh_n, dtype=np.float32)) if sh_n is not None else self.n)
        self.sh_frame.s = np.atleast_2d(np.asarray(sh_s, dtype=np.float32)) if sh_s is not None else None


class BSDF(BSDFBase):
    """mi.BSDF: sample / eval / pdf / eval_pdf with Mitsuba's signatures (CustomBSDF.py:87,177-184)."""
    _quirks = _capi.USQ_REFERENCE

    def _material(self):
        t, p = self.to_material()
        return _capi.make_material(t, p)

    def sample(self, ctx, si, sample1, sample2, active=True):
        n = len(si.wi)
        wi, ng, ns = _soa3(si.wi), _soa3(si.n, n), _soa3(si.sh_frame.n, n)
        s1 = _vec(sample1, n)
        s2a = np.asarray(sample2, dtype=np.float32)
        if s2a.ndim == 2 and s2a.shape[-1] == 2:
            s2 = _capi.f32(s2a.T)
        else:  # the reference passes a scalar Float for sample2 (CustomIntegrator.py:338)
            s2 = _capi.f32(np.stack([_vec(s2a, n), _vec(s2a, n)]))
        wo = np.empty((3, n), np.float32)
        weight = np.empty((3, n), np.float32)
        pdf = np.empty(n, np.float32)
        lobe = np.empty(n, np.uint32)
        cx = _capi.default_context()
        m = self._material()
        shs = _soa3(si.sh_frame.s, n) if getattr(si.sh_frame, "s", None) is not None else None
        cx.check(cx.lib.pbrt_bsdf_sample(cx.handle, C.byref(m), int(self._quirks), n, _capi.addr(wi), _capi.addr(ng),
                                         _capi.addr(ns), _capi.addr(shs) if shs is not None else None, _capi.addr(s1),
                                         _capi.addr(s2), _capi.addr(wo), _capi.addr(pdf), _capi.addr(weight),
                                         _capi.addr(lobe)), "pbrt_bsdf_sample")
        bs = BSDFSample3f(wo.T.copy(), pdf, np.ones(n, np.float32), self._sampled_type(lobe), lobe)
        return bs, self._weight_out(weight)

    def _sampled_type(self, lobe):
        return lobe

    def _weight_out(self, weight):
        return weight.T.copy()

    def eval_pdf(self, ctx, si, wo, active=True):
        n = len(si.wi)
        wi, wos = _soa3(si.wi), _soa3(wo, n)
        f = np.empty((3, n), np.float32)
        pdf = np.empty(n, np.float32)
        cx = _capi.default_context()
        m = self._material()
        cx.check(cx.lib.pbrt_bsdf_eval_pdf(cx.handle, C.byref(m), n, _capi.addr(wi), _capi.addr(wos), _capi.addr(f),
                                           _capi.addr(pdf)), "pbrt_bsdf_eval_pdf")
        return f.T.copy(), pdf

    def eval(self, ctx, si, wo, active=True):
        return self.eval_pdf(ctx, si, wo, active)[0]

    def pdf(self, ctx, si, wo, active=True):
        return self.eval_pdf(ctx, si, wo, active)[1]


class DiffuseBSDF(BSDF):
    def __init__(self, props):
        super().__init__(props)
        self.reflectance = rgb3(props.get("reflectance", 0.5))

    def to_material(self):
        return _capi.MAT_DIFFUSE, list(self.reflectance)

    def traverse(self, cb):
        cb.put_parameter("reflectance", self.reflectance, ParamFlags.Differentiable)


class ConductorBSDF(BSDF):
    """Mitsuba 'conductor' with no material preset: eta = 0, k = 1, a perfect mirror."""

    def __init__(self, props):
        super().__init__(props)
        mat = props.get("material", "none")
        if mat != "none" or props.has_property("eta") or props.has_property("k"):
            raise NotImplementedError("conductor: only the default perfect mirror (material 'none') is supported")
        self.specular_reflectance = rgb3(props.get("specular_reflectance", 1.0))

    def to_material(self):
        return _capi.MAT_CONDUCTOR, list(self.specular_reflectance)

    def traverse(self, cb):
        cb.put_parameter("specular_reflectance", self.specular_reflectance, ParamFlags.Differentiable)


_IOR = {"vacuum": 1.0, "air": 1.000277, "water": 1.3330, "bk7": 1.5046, "diamond": 2.419, "glass": 1.5046}


def _ior(v):
    return _IOR[v] if isinstance(v, str) else float(v)


class DielectricBSDF(BSDF):
    def __init__(self, props):
        super().__init__(props)
        self.int_ior = _ior(props.get("int_ior", "bk7"))
        self.ext_ior = _ior(props.get("ext_ior", "air"))
        self.eta = self.int_ior / self.ext_ior

    def to_material(self):
        return _capi.MAT_DIELECTRIC, [self.eta]

    def traverse(self, cb):
        cb.put_parameter("eta", self.eta, ParamFlags.Differentiable)


class UltraBSDF(BSDF):
    """Acoustic impedance interface with GGX micro-facets (CustomBSDF.py:7-191).
    props: impedance (1.54), roughness (0.5)  (CustomBSDF.py:12-18)
    sample() returns (bs, acoustic_response) like CustomBSDF.py:175; eval/pdf/eval_pdf return 0
    (CustomBSDF.py:177-184).  `quirks` selects literal reference behaviour vs intent
    (include/pbrt_hip.h PBRT_USQ_*; default: the literal reference arithmetic)."""
    medium_z = 1.2  # CustomBSDF.py:105

    def __init__(self, props):
        super().__init__(props)
        self.impedance = 1.54
        if props.has_property("impedance"):
            self.impedance = float(props["impedance"])
        self.roughness = 0.5
        if props.has_property("roughness"):
            self.roughness = float(props["roughness"])
        self._quirks = int(props.get("quirks", _capi.USQ_REFERENCE))
        refl = BSDFFlags.DeltaReflection | BSDFFlags.FrontSide | BSDFFlags.BackSide
        trans = BSDFFlags.DeltaTransmission | BSDFFlags.FrontSide | BSDFFlags.BackSide
        self.m_components = [refl, trans]
        self.m_flags = refl | trans

    def to_material(self):
        return _capi.MAT_ULTRA, [float(self.impedance), float(self.roughness), self.medium_z]

    def _sampled_type(self, lobe):
        return np.where(lobe == 0, BSDFFlags.GlossyReflection, BSDFFlags.GlossyTransmission).astype(np.uint32)

    def _weight_out(self, weight):
        return weight[0].copy()  # scalar acoustic amplitude (CustomBSDF.py:170-175)

    def eval(self, ctx, si, wo, active=True):
        return 0.0

    def pdf(self, ctx, si, wo, active=True):
        return 0.0

    def eval_pdf(self, ctx, si, wo, active=True):
        return 0.0, 0.0

    def traverse(self, callback):
        callback.put_parameter("impedance", self.impedance, ParamFlags.Differentiable)
        callback.put_parameter("roughness", self.roughness, ParamFlags.Differentiable)

    def parameters_changed(self, keys=None):
        self.impedance = float(np.asarray(self.impedance).ravel()[0])
        self.roughness = float(np.asarray(self.roughness).ravel()[0])


# ================================================================================================
# Emitters
# ================================================================================================
class Emitter(EmitterBase):
    pass


class AreaEmitter(Emitter):
    """Mitsuba 'area' emitter attached to a shape.  [DEFINE] 'ultraray' on the cbox luminaire
    (scenes/cbox.xml:64-84, registered nowhere in the reference) is treated as an area emitter
    whose radiance is its 'intensity' (1,1,1)."""

    def __init__(self, props):
        super().__init__(props)
        v = props.get("radiance", None)
        if v is None:
            v = props.get("intensity", 1.0)
        self.radiance = rgb3(v)

    def radiance_rgb(self):
        return np.asarray(self.radiance, dtype=np.float64)

    def traverse(self, cb):
        cb.put_parameter("radiance", self.radiance, ParamFlags.Differentiable)


class PointEmitter(Emitter):
    def __init__(self, props):
        super().__init__(props)
        self.intensity = rgb3(props.get("intensity", 1.0))
        pos = props.get("position", None)
        tw = props.get("to_world", None)
        if pos is None:
            pos = (tw.translation() if tw is not None else np.zeros(3))
        self.position = np.asarray(pos, dtype=np.float64)

    def device_emitter(self):
        return _capi.EMIT_POINT, self.intensity, self.position

    def traverse(self, cb):
        cb.put_parameter("intensity", self.intensity, ParamFlags.Differentiable)
        cb.put_parameter("position", self.position, ParamFlags.Differentiable)


class EmitterFlags:
    Surface = 0x10
    SpatiallyVarying = 0x20


class CustomEmitter(Emitter):
    """Transducer array as a ray source (CustomEmmitter.py:5-129): same props, same defaults;
    compute_element_geometry / sample_position / sample_ray / sample_ray_differential.
    (The reference crashes in its constructor on a typo, CustomEmmitter.py:25; here the
    correctly spelled method is called.)  Not part of the radiance light list: the integrator
    never calls it."""
    _param_attr = {"rays_per_element": "number_of_rays_per_element"}
    is_transducer = True  # a source of acoustic rays, not a light of the radiance scene (scene.flatten skips it)

    def __init__(self, props):
        super().__init__(props)
        self.number_of_elements = int(props.get("number_of_elements", 64))
        self.pitch = float(props.get("pitch", 0.0003))
        self.element_width = float(props.get("element_width", 0.0003))
        self.element_height = float(props.get("element_height", 0.0005))
        self.radius = float(props.get("radius", 0.0))
        self.opening_angle = float(props.get("opening_angle", 0.0))
        self.number_of_rays_per_element = int(props.get("number_of_rays_per_element", 1))
        self.number_of_total_rays = self.number_of_elements * self.number_of_rays_per_element
        self.speed_of_sound = float(props.get("speed_of_sound", 1540))
        self.steering_angle_min = float(props.get("steering_angle_min", -10.0))
        self.steering_angle_max = float(props.get("steering_angle_max", 10.0))
        self.element_positions, self.element_normals = self.compute_element_geometry()
        self._flags = EmitterFlags.Surface | EmitterFlags.SpatiallyVarying
        self._id = props.id()

    def device_emitter(self):
        raise NotImplementedError("CustomEmitter is a transducer ray source, not a radiance emitter")

    def compute_element_geometry(self):
        n = self.number_of_elements
        if self.radius == 0.0:  # CustomEmmitter.py:33-38
            x = np.linspace(-(n - 1) / 2 * self.pitch, (n - 1) / 2 * self.pitch, n, dtype=np.float32)
            pos = np.stack([x, np.zeros_like(x), np.zeros_like(x)], axis=1)
            nrm = np.tile(np.array([[0, 0, 1]], np.float32), (n, 1))
        else:  # :41-47
            span = math.radians(self.opening_angle)
            th = np.linspace(-span / 2, span / 2, n, dtype=np.float32)
            pos = np.stack([self.radius * np.sin(th), np.zeros_like(th), self.radius * np.cos(th)], axis=1)
            nrm = np.stack([np.sin(th), np.zeros_like(th), np.cos(th)], axis=1)
        return pos.astype(np.float32), nrm.astype(np.float32)

    def _desc(self):
        e = _capi.UsEmitter()
        e.number_of_elements = self.number_of_elements
        e.pitch, e.element_width, e.element_height = self.pitch, self.element_width, self.element_height
        e.radius, e.opening_angle = self.radius, self.opening_angle
        e.number_of_rays_per_element = self.number_of_rays_per_element
        e.speed_of_sound = self.speed_of_sound
        e.steering_angle_min, e.steering_angle_max = self.steering_angle_min, self.steering_angle_max
        return e

    def _call(self, time, sample1, sample2, sample3):
        s2 = np.atleast_2d(np.asarray(sample2, dtype=np.float32))
        n = len(s2)
        t, s1, s3 = _vec(time, n), _vec(sample1, n), _vec(sample3, n)
        s2s = _capi.f32(s2.T)
        o = np.empty((3, n), np.float32)
        d = np.empty((3, n), np.float32)
        rt = np.empty(n, np.float32)
        w = np.empty(n, np.float32)
        pdf = np.empty(n, np.float32)
        cx = _capi.default_context()
        e = self._desc()
        cx.check(cx.lib.pbrt_us_emitter_sample_ray(cx.handle, C.byref(e), n, _capi.addr(t), _capi.addr(s1),
                                                   _capi.addr(s2s), _capi.addr(s3), _capi.addr(o), _capi.addr(d),
                                                   _capi.addr(rt), _capi.addr(w), _capi.addr(pdf)),
                 "pbrt_us_emitter_sample_ray")
        return o.T.copy(), d.T.copy(), rt, w, pdf

    def sample_position(self, time, sample, active=True):
        """-> (PositionSample3f-like dict(p, n, time, delta), pdf)   (CustomEmmitter.py:51-79)"""
        sample1, sample2 = sample
        o, _, _, _, pdf = self._call(time, sample1, sample2, 0.0)
        n = len(o)
        idx = np.minimum(np.floor(_vec(sample1, n) * self.number_of_elements), self.number_of_elements - 1).astype(int)
        return dict(p=o, n=self.element_normals[idx], time=_vec(time, n), delta=False), pdf

    def sample_ray(self, time, sample1, sample2, sample3, active=True):
        """-> (Ray3f-like dict(o, d, time), weight)   (CustomEmmitter.py:81-107)"""
        o, d, rt, w, _ = self._call(time, sample1, sample2, sample3)
        return dict(o=o, d=d, time=rt), w

    def sample_ray_differential(self, *args, **kwargs):
        ray, spec = self.sample_ray(*args, **kwargs)
        return ray, spec, None

    def traverse(self, callback):  # CustomEmmitter.py:114-124
        for k in ("number_of_elements", "pitch", "element_width", "element_height", "radius", "opening_angle",
                  "steering_angle_min", "steering_angle_max", "speed_of_sound"):
            callback.put_parameter(k, getattr(self, k), ParamFlags.Differentiable)
        callback.put_parameter("rays_per_element", self.number_of_rays_per_element, ParamFlags.Differentiable)

    def parameters_changed(self, keys=None):  # CustomEmmitter.py:126-129
        self.element_positions, self.element_normals = self.compute_element_geometry()
        self.number_of_total_rays = self.number_of_elements * self.number_of_rays_per_element


# ================================================================================================
# Sensors
# ================================================================================================
class Sensor(SensorBase):
    def __init__(self, props):
        super().__init__(props)
        tw = props.get("to_world", None)
        if tw is not None and not isinstance(tw, ScalarTransform4f):
            tw = ScalarTransform4f(getattr(tw, "matrix", tw))
        self.transform = tw if tw is not None else ScalarTransform4f()
        self._film, self._sampler = None, None
        for k in props.property_names():
            v = props.get(k)
            if isinstance(v, Film):
                self._film = v
            elif isinstance(v, Sampler):
                self._sampler = v

    def film(self):
        return self._film

    def sampler(self):
        return self._sampler

    def world_transform(self):
        return self.transform

    def _children(self):
        out = []
        if self._film is not None:
            out.append(("film", self._film))
        if self._sampler is not None:
            out.append(("sampler", self._sampler))
        return out


class PerspectiveSensor(Sensor):
    """Mitsuba 'perspective' (scenes/cbox.xml:11-32, scenes/simple.xml:7-21)."""

    def __init__(self, props):
        super().__init__(props)
        if self._film is None:
            self._film = Film(Properties("hdrfilm"))
        if self._sampler is None:
            self._sampler = Sampler(Properties("independent"))
        self.near_clip = float(props.get("near_clip", 1e-2))
        self.far_clip = float(props.get("far_clip", 1e4))
        w, h = self._film.size()
        aspect = w / h
        if props.has_property("fov"):
            fov = float(props["fov"])
            axis = props.get("fov_axis", "x")
        else:  # focal_length default "50mm" on 36x24 mm film, diagonal axis
            f = props.get("focal_length", "50mm")
            f = float(str(f).replace("mm", ""))
            fov = math.degrees(2.0 * math.atan(math.sqrt(36.0 ** 2 + 24.0 ** 2) / (2.0 * f)))
            axis = "diagonal"
        if axis == "smaller":
            axis = "y" if aspect > 1 else "x"
        elif axis == "larger":
            axis = "x" if aspect > 1 else "y"
        if axis == "x":
            fx = fov
        elif axis == "y":
            fx = math.degrees(2.0 * math.atan(math.tan(math.radians(fov) / 2.0) * aspect))
        elif axis == "diagonal":
            diag = 2.0 * math.tan(math.radians(fov) / 2.0)
            width = diag / math.sqrt(1.0 + 1.0 / (aspect * aspect))
            fx = math.degrees(2.0 * math.atan(width / 2.0))
        else:
            raise ValueError(f"unknown fov_axis {axis}")
        self.x_fov = fx

    def camera(self) -> _capi.Camera:
        cam = _capi.Camera()
        cam.to_world = _capi.mat12(self.transform.matrix)
        cam.tan_half_fov_x = math.tan(math.radians(self.x_fov) / 2.0)
        cam.near_clip, cam.far_clip = self.near_clip, self.far_clip
        cam.film_w, cam.film_h = self._film.size()
        return cam

    def sample_ray(self, time, wavelength_sample, position_sample, aperture_sample, active=True):
        """position_sample [n,2] in [0,1)^2 -> (dict(o, d, maxt), weight)"""
        pos = np.atleast_2d(np.asarray(position_sample, dtype=np.float32))
        n = len(pos)
        ps = _capi.f32(pos.T)
        o = np.empty((3, n), np.float32)
        d = np.empty((3, n), np.float32)
        tmax = np.empty(n, np.float32)
        cx = _capi.default_context()
        cam = self.camera()
        cx.check(cx.lib.pbrt_sensor_sample_ray(cx.handle, C.byref(cam), n, _capi.addr(ps), _capi.addr(o), _capi.addr(d),
                                               _capi.addr(tmax)), "pbrt_sensor_sample_ray")
        return dict(o=o.T.copy(), d=d.T.copy(), maxt=tmax), np.ones(n, np.float32)

    def traverse(self, cb):
        cb.put_parameter("x_fov", self.x_fov, ParamFlags.NonDifferentiable)


class UltraSensor(Sensor):
    """Transducer as a Mitsuba sensor (class recovered from stale bytecode; this is
    what USMain.py:17-18 imports and MitsubaScenes/*.xml:16-34 configure).  The integrator only reads
    `.transform` (CustomIntegrator.py:272)."""

    def __init__(self, props):
        super().__init__(props)
        self.num_elements_lateral = int(props.get("num_elements_lateral", 128))
        self.element_width = float(props.get("elements_width", 0.003))
        self.element_height = float(props.get("elements_height", 0.01))
        self.pitch = float(props.get("pitch", 0.00035))
        self.radius = float(props.get("radius", math.inf))
        self.center_frequency = float(props.get("center_frequency", 5e6))
        self.sound_speed = float(props.get("sound_speed", 1540))
        self.emission_time = 0.0
        self.directivity = float(props.get("directivity", 1.0))

    def _desc(self):
        s = _capi.UsSensor()
        s.num_elements = self.num_elements_lateral
        s.element_width, s.element_height, s.pitch = self.element_width, self.element_height, self.pitch
        s.radius = self.radius
        s.center_frequency, s.sound_speed, s.directivity = self.center_frequency, self.sound_speed, self.directivity
        s.to_world = _capi.mat12(self.transform.matrix)
        return s

    def sample_ray(self, time, wavelength_sample, position_sample, aperture_sample, active=True,
                   use_hemisphere_warp=True):
        pos = np.atleast_2d(np.asarray(position_sample, dtype=np.float32))
        ap = np.atleast_2d(np.asarray(aperture_sample, dtype=np.float32))
        n = len(pos)
        t, wl = _vec(time, n), _vec(wavelength_sample, n)
        ps, aps = _capi.f32(pos.T), _capi.f32(np.broadcast_to(ap, (n, 2)).T)
        o = np.empty((3, n), np.float32)
        d = np.empty((3, n), np.float32)
        w = np.empty(n, np.float32)
        cx = _capi.default_context()
        s = self._desc()
        cx.check(cx.lib.pbrt_us_sensor_sample_ray(cx.handle, C.byref(s), int(bool(use_hemisphere_warp)), n,
                                                  _capi.addr(t), _capi.addr(wl), _capi.addr(ps), _capi.addr(aps),
                                                  _capi.addr(o), _capi.addr(d), _capi.addr(w)),
                 "pbrt_us_sensor_sample_ray")
        return dict(o=o.T.copy(), d=d.T.copy()), w


class CustomSensor(Sensor):
    """Receive-side accumulator (CustomSensor.py:7-76): put_data / channel_data / clear."""

    def __init__(self, props):
        super().__init__(props)
        self.number_of_elements = int(props.get("number_of_elements", 128))
        self.pitch = float(props.get("pitch", 0.0003))
        self.element_width = float(props.get("element_width", 0.00027))
        self.element_height = float(props.get("element_height", 0.005))
        self.sample_rate = float(props.get("sample_rate", 50e6))
        self.speed_of_sound = float(props.get("speed_of_sound", 1540.0))
        self.time_samples = int(props.get("time_samples", 3000))
        self.channel_buffer = np.zeros((self.number_of_elements, self.time_samples), dtype=np.float32)

    def put_data(self, ray, amplitude, active=True):
        """ray: dict(o [n,3], d [n,3], time [n]) (or a single ray); amplitude [n]   (CustomSensor.py:29-59)"""
        o = np.atleast_2d(np.asarray(ray["o"], dtype=np.float32))
        d = np.atleast_2d(np.asarray(ray["d"], dtype=np.float32))
        n = len(o)
        t, amp = _vec(ray["time"], n), _vec(amplitude, n)
        ox, ds = _capi.f32(o[:, 0]), _capi.f32(d.T)
        r = _capi.UsReceiver()
        r.number_of_elements, r.pitch, r.sample_rate, r.time_samples = (self.number_of_elements, self.pitch,
                                                                         self.sample_rate, self.time_samples)
        cx = _capi.default_context()
        cx.check(cx.lib.pbrt_us_put_data(cx.handle, C.byref(r), n, _capi.addr(ox), _capi.addr(t), _capi.addr(ds),
                                         _capi.addr(amp), _capi.addr(self.channel_buffer)), "pbrt_us_put_data")

    def channel_data(self):
        return self.channel_buffer

    def clear(self):
        self.channel_buffer = np.zeros((self.number_of_elements, self.time_samples), dtype=np.float32)

    def traverse(self, callback):  # CustomSensor.py:67-73
        for k in ("number_of_elements", "pitch", "element_width", "element_height", "sample_rate", "speed_of_sound"):
            callback.put_parameter(k, getattr(self, k), ParamFlags.NonDifferentiable)

    def parameters_changed(self, keys=None):  # CustomSensor.py:75-76 ('parameters')
        self.clear()

    parameters = parameters_changed


# ================================================================================================
# Integrators
# ================================================================================================
class SamplingIntegrator(IntegratorBase):
    """mi.SamplingIntegrator: render(scene, sensor, seed, spp) + sample(scene, sampler, ray, medium, active)."""
    max_depth = 0xFFFFFFFF
    rr_depth = 5

    def _film_desc(self, scene, sensor, seed, spp, crop=None, sample_offset=0, raw=False, pass_paths=0, flags=0):
        film = sensor.film()
        fd = _capi.FilmDesc()
        cx, cy, cw, ch = crop if crop is not None else film.crop
        fd.crop_x, fd.crop_y, fd.crop_w, fd.crop_h = int(cx), int(cy), int(cw), int(ch)
        fd.spp = int(spp) if spp else sensor.sampler().sample_count
        fd.sample_offset = int(sample_offset)
        fd.max_depth = min(int(self.max_depth) if self.max_depth >= 0 else 0xFFFFFFFF, 0xFFFFFFFF)
        fd.rr_depth = int(self.rr_depth)
        fd.filter = film.rfilter.kind
        fd.seed = int(seed) & 0xFFFFFFFF
        fd.flags = (_capi.FILM_RAW_ACCUM if raw else 0) | int(flags)
        fd.pass_paths = int(pass_paths)
        return fd

    def render(self, scene, sensor=0, seed=0, spp=0, crop=None, sample_offset=0, raw=False, out_dev=None, pass_paths=0,
               flags=0):
        """-> float32 [crop_h, crop_w, 3] (4 with raw=True).  With out_dev (a device pointer, e.g.
        tensor.data_ptr()) the film stays in HBM and None is returned."""
        sens = scene.sensors()[sensor] if isinstance(sensor, int) else sensor
        if not isinstance(sens, PerspectiveSensor):
            raise TypeError("radiance rendering needs a 'perspective' sensor")
        fd = self._film_desc(scene, sens, seed, spp, crop, sample_offset, raw, pass_paths, flags)
        cam = sens.camera()
        dev = scene.device()
        nch = 4 if raw else 3
        if out_dev is not None:
            dev.ctx.check(dev.ctx.lib.pbrt_render_radiance_dev(dev.handle, C.byref(cam), C.byref(fd), C.c_void_p(int(out_dev))),
                          "pbrt_render_radiance_dev")
            return None
        out = np.empty((fd.crop_h, fd.crop_w, nch), np.float32)
        dev.ctx.check(dev.ctx.lib.pbrt_render_radiance(dev.handle, C.byref(cam), C.byref(fd), _capi.addr(out)),
                      "pbrt_render_radiance")
        return out

    def sample(self, scene, sampler, ray, medium=None, active=True):
        """Radiance along a batch of rays: ray = dict(o [n,3], d [n,3], maxt [n] optional).
        -> (rgb [n,3], valid mask [n], aovs [])
        Ray i draws its random numbers from the key (sampler.index_offset + i, sampler.sample_index) under sampler.seed
        (all 0 without a sampler): rays listed in pixel order and generated with a render's jitter ARE that render's
        paths, so rgb[i] is sample `sample_index` of pixel i (box filter, 1 spp: the film itself)."""
        o = np.atleast_2d(np.asarray(ray["o"], dtype=np.float32))
        d = np.atleast_2d(np.asarray(ray["d"], dtype=np.float32))
        n = len(o)
        tm = _vec(ray.get("maxt", np.inf), n)
        os_, ds_ = _capi.f32(o.T), _capi.f32(d.T)
        rgb = np.empty((3, n), np.float32)
        seed = getattr(sampler, "seed", 0) if sampler is not None else 0
        sidx = getattr(sampler, "sample_index", 0) if sampler is not None else 0
        ioff = getattr(sampler, "index_offset", 0) if sampler is not None else 0
        dev = scene.device()
        md = min(int(self.max_depth) if self.max_depth >= 0 else 0xFFFFFFFF, 0xFFFFFFFF)
        dev.ctx.check(dev.ctx.lib.pbrt_integrator_sample(dev.handle, n, _capi.addr(os_), _capi.addr(ds_), _capi.addr(tm),
                                                          int(ioff), int(sidx), int(seed) & 0xFFFFFFFF, md, int(self.rr_depth),
                                                          _capi.addr(rgb)), "pbrt_integrator_sample")
        return rgb.T.copy(), np.ones(n, bool), []


class PathIntegrator(SamplingIntegrator):
    """Mitsuba 'path' (scenes/cbox.xml:5-9): max_depth (-1 = unbounded), rr_depth 5."""

    def __init__(self, props):
        super().__init__(props)
        md = int(props.get("max_depth", -1))
        self.max_depth = md if md >= 0 else 0xFFFFFFFF
        self.rr_depth = int(props.get("rr_depth", 5))
        self.hide_emitters = bool(props.get("hide_emitters", False))

    def traverse(self, cb):
        cb.put_parameter("max_depth", self.max_depth, ParamFlags.NonDifferentiable)


class DirectIntegrator(SamplingIntegrator):
    """Mitsuba 'direct' (scenes/simple.xml:5) with emitter_samples = bsdf_samples = 1: identical
    estimator to 'path' with max_depth 2 (one emitter-sampling and one BSDF-sampling strategy
    combined by MIS)."""

    def __init__(self, props):
        super().__init__(props)
        if int(props.get("emitter_samples", 1)) != 1 or int(props.get("bsdf_samples", 1)) != 1 or \
                int(props.get("shading_samples", 1)) != 1:
            raise NotImplementedError("direct: only emitter_samples = bsdf_samples = 1")
        self.max_depth = 2
        self.rr_depth = 5


class UltraIntegrator(SamplingIntegrator):
    """Ultrasound acquisition integrator (CustomIntegrator.py:12-412): same props, defaults and
    attributes; simulate_acquisition_parallel(scene) / simulate_acquisition(scene) run the whole
    (angle, element, path) job on the GPU and leave `channel_buf` (n_angles, n_elements,
    time_samples) float32 and `transmission_delays_buf` behind, as USMain.py:103-121 expects.

    Extra props (not in the reference): paths_per_ray (1) independent Monte-Carlo paths per
    (angle, element) primary ray, seed (0), quirks (PBRT_USQ_REFERENCE)."""

    def __init__(self, props):
        super().__init__(props)
        self.max_depth = int(props.get("max_depth", 2))
        self.frequency = float(props.get("frequency", 5e6))
        self.sound_speed = float(props.get("sound_speed", 1540))
        self.attenuation = float(props.get("attenuation", 0.5))
        self.wave_cycles = props.get("wave_cycles", 5)  # read, unused (CustomIntegrator.py:20)
        self.main_beam_angle = float(props.get("main_beam_angle", 10))
        self.cutoff_angle = float(props.get("cutoff_angle", 20))
        self.fs = float(props.get("sampling_rate", 50e6))
        self.n_elements = int(props.get("n_elements", 128))
        self.pitch = float(props.get("pitch", 0.00035))
        self.elem_x = as_dr(self.pitch * (np.arange(self.n_elements, dtype=np.float32) - (self.n_elements - 1) / 2))
        self.trans_norm = np.array([0.0, 0.0, 1.0], np.float32)
        ang = props.get("angles", None)
        if ang is None:
            ang = np.linspace(-30, 30, 25, dtype=np.float32)
        elif isinstance(ang, str):
            ang = [float(t) for t in ang.replace(",", " ").split()]
        self.angles = as_dr(np.asarray(ang, dtype=np.float32).ravel())
        self.n_angles = len(self.angles)
        self.init_amp, self.init_atten, self.init_tof = 1.0, 1.0, 0.0
        self.time_samples = int(props.get("time_samples", 3000))
        self.channel_buf = np.zeros(self.n_angles * self.n_elements * self.time_samples, np.float32)
        self.transmission_delays_buf = np.zeros(self.n_angles * self.n_elements, np.float32)
        self.ray_count = 0
        self.paths_per_ray = int(props.get("paths_per_ray", 1))
        self.seed = int(props.get("seed", 0))
        self.quirks = int(props.get("quirks", _capi.USQ_REFERENCE))
        # SURVEY f-3: 'impulse' = the reference's one-sample p * sin(phase) (CustomIntegrator.py:348-354); 'gaussian' = the
        # pulse of the prototype (RayTracingV0.py:194-204), length set by the otherwise unused wave_cycles:
        # sigma = wave_cycles / (4 f)  [DEFINE]
        self.pulse_model = str(props.get("pulse_model", "impulse"))
        if self.pulse_model not in ("impulse", "gaussian"):
            raise ValueError("pulse_model must be 'impulse' or 'gaussian'")
        if self.pulse_model == "gaussian":
            self.quirks |= _capi.USQ_NO_CARRIER
        self.max_path_len = 0.2  # CustomIntegrator.py:307,372
        # Where a path's primary ray comes from: "element" = the integrator's own deterministic ray (CustomIntegrator.py:264-273);
        # "emitter" = every path draws its ray from the scene's CustomEmitter.sample_ray (CustomEmmitter.py:81-107), the
        # (angle, element) grid stratifying the emitter's element pick and steering range  [DEFINE, DESIGN D15: the reference
        # never connects the two classes; BASELINE config 3 names them together]
        self.primary_rays = str(props.get("primary_rays", "element"))
        if self.primary_rays not in ("element", "emitter"):
            raise ValueError("primary_rays must be 'element' or 'emitter'")

    # ray_count (CustomIntegrator.py:231,360,402): segments traced by the last acquisition; after a queued acquisition the counter
    # is fetched when somebody reads it (that waits for the stream)
    @property
    def ray_count(self):
        if self._ray_count is None and getattr(self, "_stats_ctx", None) is not None:
            self._ray_count = int(self._stats_ctx.stats()["segments"])
        return self._ray_count or 0

    @ray_count.setter
    def ray_count(self, value):
        self._ray_count = int(value)

    # channel_buf (CustomIntegrator.py:43,260; read at USMain.py:103): a host array, as in the reference.  When an acquisition
    # left its result in HBM (us_render keeps the whole loop on the device) the copy to the host happens on first read.
    @property
    def channel_buf(self):
        if self._channel_host is None and self._channel_dev is not None:
            self._channel_host = self._channel_dev.numpy().reshape(self.n_angles, self.n_elements, self.time_samples)
        return self._channel_host

    @channel_buf.setter
    def channel_buf(self, value):
        self._channel_host = value
        self._channel_dev = None

    def _set_device_channel(self, dev_buffer):
        """the channel buffer of the last acquisition sits in this DeviceBuffer; channel_buf fetches it when read"""
        self._channel_host = None
        self._channel_dev = dev_buffer

    def sample(self, scene, sampler, ray, medium=None, active=True):  # CustomIntegrator.py:52-53
        n = len(np.atleast_2d(np.asarray(ray["o"]))) if isinstance(ray, dict) else 1
        return np.zeros(n, np.float32), active, []

    @property
    def pulse_sigma(self) -> float:
        """Gaussian width of the 'gaussian' pulse model: wave_cycles / (4 f) seconds  [DEFINE, SURVEY f-3]"""
        return float(self.wave_cycles) / (4.0 * self.frequency)

    def us_params(self, scene, quirks=None) -> _capi.UsParams:
        if self.n_angles > _capi.US_MAX_ANGLES:
            raise ValueError(f"at most {_capi.US_MAX_ANGLES} plane-wave angles")
        p = _capi.UsParams()
        p.max_depth = self.max_depth
        p.frequency, p.sound_speed, p.attenuation = self.frequency, self.sound_speed, self.attenuation
        p.main_beam_angle, p.cutoff_angle, p.fs = self.main_beam_angle, self.cutoff_angle, self.fs
        p.n_elements, p.pitch, p.n_angles = self.n_elements, self.pitch, self.n_angles
        for i, a in enumerate(np.asarray(self.angles, dtype=np.float32)):
            p.angles_deg[i] = float(a)
        p.time_samples = self.time_samples
        sens = scene.sensors()[0] if scene is not None and scene.sensors() else None
        T = sens.transform.matrix if sens is not None else np.eye(4)  # CustomIntegrator.py:272
        p.sensor_to_world = _capi.mat12(T)
        p.max_path_len = self.max_path_len
        p.quirks = int(self.quirks if quirks is None else quirks)
        p.primary = _capi.US_PRIMARY_ELEMENT
        if self.primary_rays == "emitter":
            ems = [e for e in (scene.emitters() if scene is not None else []) if getattr(e, "is_transducer", False)]
            if not ems:
                raise ValueError("primary_rays='emitter' needs an 'ultrasound_emitter' (CustomEmitter) in the scene")
            if ems[0].number_of_elements != self.n_elements:
                raise ValueError(f"the emitter has {ems[0].number_of_elements} elements, the integrator {self.n_elements}: "
                                 "the acquisition grid stratifies the emitter's elements, they must be the same array")
            p.primary = _capi.US_PRIMARY_EMITTER
            p.emitter = ems[0]._desc()
        return p

    def _acquire(self, scene, quirks, paths_per_ray=None, path_offset=0, norm_paths=None, seed=None, out_dev=None, pulse=None,
                 queue=False):
        """pulse: apply the Gaussian-windowed carrier when the echoes were deposited without one (pulse_model
        'gaussian' = PBRT_USQ_NO_CARRIER).  Default: yes for a host buffer.  A device buffer (out_dev) is one shard of a
        sum that is still to be reduced, so the caller convolves the REDUCED buffer once (parallel.distributed_acquire
        does) and says pulse=False here; leaving it unsaid is an error rather than a silently carrier-less result."""
        ppr = int(paths_per_ray if paths_per_ray is not None else self.paths_per_ray)
        norm = int(norm_paths if norm_paths is not None else ppr)
        p = self.us_params(scene, quirks)
        dev = scene.device()
        tx = np.empty(self.n_angles * self.n_elements, np.float32)
        sd = int(self.seed if seed is None else seed) & 0xFFFFFFFF
        no_carrier = bool(int(p.quirks) & _capi.USQ_NO_CARRIER)
        if out_dev is not None and no_carrier and pulse is not False:
            raise ValueError("pulse_model 'gaussian' with a device output buffer: the pulse is applied to the reduced buffer "
                             "(parallel.distributed_acquire); pass pulse=False to get this shard's bare echo amplitudes")
        if out_dev is not None:
            # queue=True: pbrt_us_acquire_queue_dev -- the call returns with the acquisition queued on the context's stream; its
            # statistics (ray_count) arrive with the next call that waits
            fn = dev.ctx.lib.pbrt_us_acquire_queue_dev if queue else dev.ctx.lib.pbrt_us_acquire_dev
            dev.ctx.check(fn(dev.handle, C.byref(p), sd, ppr, int(path_offset), norm, C.c_void_p(int(out_dev)), _capi.addr(tx)),
                          "pbrt_us_acquire_queue_dev" if queue else "pbrt_us_acquire_dev")
            buf = None
        else:
            buf = np.empty((self.n_angles, self.n_elements, self.time_samples), np.float32)
            dev.ctx.check(dev.ctx.lib.pbrt_us_acquire(dev.handle, C.byref(p), sd, ppr, int(path_offset), norm,
                                                       _capi.addr(buf), _capi.addr(tx)), "pbrt_us_acquire")
        self.transmission_delays_buf = tx
        self._ray_count = None if queue else int(dev.ctx.stats()["segments"])
        self._stats_ctx = dev.ctx
        if buf is not None and no_carrier and pulse is not False:
            # f-3 pulse model: the echoes were deposited as plain amplitudes; give every trace the Gaussian-windowed carrier
            from .beamform import apply_pulse
            buf = apply_pulse(buf, self.fs, self.frequency, self.pulse_sigma)
        return buf

    def simulate_acquisition_parallel(self, scene):  # CustomIntegrator.py:235-405
        self.channel_buf = self._acquire(scene, self.quirks)
        return True

    def simulate_acquisition(self, scene):  # CustomIntegrator.py:60-232 (Dr.Jit variant: App. B -- B1 frozen draws, B2, B3, B5)
        q = self.quirks | _capi.USQ_DRJIT_VARIANT
        self.channel_buf = self._acquire(scene, q).reshape(-1)
        return True

    def traverse(self, callback):  # CustomIntegrator.py:408-409
        callback.put_parameter("pitch", self.pitch, ParamFlags.Differentiable)

    def parameters_changed(self, keys=None):
        self.elem_x = as_dr(self.pitch * (np.arange(self.n_elements, dtype=np.float32) - (self.n_elements - 1) / 2))


# ------------------------------------------------------------------------------------------------
# registration of the built-ins and of the reference's plugin names (USMain.py:14-24)
# ------------------------------------------------------------------------------------------------
for _n, _c in (("path", PathIntegrator), ("direct", DirectIntegrator), ("ultrasound_integrator", UltraIntegrator)):
    _register("integrator", _n, _c)
for _n, _c in (("diffuse", DiffuseBSDF), ("conductor", ConductorBSDF), ("dielectric", DielectricBSDF),
               ("ultrasound_bsdf", UltraBSDF)):
    _register("bsdf", _n, _c)
for _n, _c in (("area", AreaEmitter), ("ultraray", AreaEmitter), ("point", PointEmitter),
               ("ultrasound_emitter", CustomEmitter)):
    _register("emitter", _n, _c)
for _n, _c in (("perspective", PerspectiveSensor), ("ultrasound_sensor", UltraSensor), ("custom_sensor", CustomSensor)):
    _register("sensor", _n, _c)
for _n, _c in (("obj", MeshShape), ("ply", MeshShape), ("sphere", SphereShape), ("rectangle", RectangleShape),
               ("cone", ConeShape)):
    _register("shape", _n, _c)
_register("film", "hdrfilm", Film)
_register("sampler", "independent", Sampler)
for _n in ("box", "tent", "gaussian"):
    _register("rfilter", _n, ReconstructionFilter)
