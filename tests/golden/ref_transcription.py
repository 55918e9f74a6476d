"""Independent float64 transcription of the reference's hot loop -- TEST INFRASTRUCTURE, pure Python / NumPy.

Purpose ("pin the restatement independently of oracle.cpp"): oracle/oracle.cpp and the HIP kernels
share their leaf arithmetic statement for statement, so "HIP == oracle" cannot see a transcription error.  This
module restates the same path a SECOND time, from the reference's Python directly and in another arithmetic
(f64, textbook operation order, libm transcendentals, no shared helper, no shared source):

  ultrasound  _trace_single_ray            /root/reference/CustomIntegrator.py:262-376  -> us_trace_single_ray
              UltraBSDF._ggx_sample/sample /root/reference/CustomBSDF.py:30-61, :87-175 -> ultra_bsdf_sample
  radiance    Mitsuba 3 `path` with diffuse / conductor / dielectric / area emitter / perspective sensor as the
              reference's scenes/cbox.xml configures them (the arithmetic lives in Mitsuba,
              which is absent here)                                                        -> path_radiance

Every Mitsuba call the reference makes is restated from Mitsuba 3's published definition and named where it is
used (Frame3f / coordinate_system, SurfaceInteraction::initialize_sh_frame, warp::square_to_uniform_disk_concentric,
Shape::ray_intersect of `rectangle` and `sphere`, SurfaceInteraction::spawn_ray / offset_p, fresnel, mis_weight).
The reference draws its random numbers from an unseeded NumPy generator (:283,319,337,365); here the draws are
INPUTS, so a fixture is (geometry, parameters, draws) -> (echo bin, pressure, pdf, amplitude, direction).

What is a build definition rather than reference behaviour, and therefore followed here so that values are
comparable sample by sample: the counter-based generator that supplies the draws (pcg4d,
`rng4` below), the repair of quirk B5 (`survive = True`), one uniform point per PARALLELOGRAM light (D10), shading
frames of the radiance BSDFs from coordinate_system(n) (D13).  tests/golden/make_golden.py runs this module and
commits its outputs as fixtures K9 (ultrasound) and K10 (radiance); tests/test_pinned_transcription.py checks the
oracle against them, tests/test_gpu_pinned.py the HIP library."""
from __future__ import annotations

import math

import numpy as np

RAY_EPSILON = 1500.0 * 2.0 ** -24      # Mitsuba math::RayEpsilon<float>
SHADOW_EPSILON = 10.0 * RAY_EPSILON    # Mitsuba math::ShadowEpsilon<float>


# --------------------------------------------------------------------------------------------------------------------
# draws: pcg4d (Jarzynski & Olano, "Hash Functions for GPU Rendering", JCGT 2020, listing of pcg4d), the build's
# counter-based generator; a uniform is the top 24 bits / 2^24
# --------------------------------------------------------------------------------------------------------------------
def pcg4d(a, b, c, d):
    M = 0xFFFFFFFF
    v = [(x * 1664525 + 1013904223) & M for x in (a, b, c, d)]
    for _ in range(2):
        v[0] = (v[0] + v[1] * v[3]) & M
        v[1] = (v[1] + v[2] * v[0]) & M
        v[2] = (v[2] + v[0] * v[1]) & M
        v[3] = (v[3] + v[1] * v[2]) & M
        if _ == 0:
            v = [x ^ (x >> 16) for x in v]
    return v


def rng4(a, b, c, seed):
    return [(x >> 8) / 16777216.0 for x in pcg4d(a, b, c, seed)]


# --------------------------------------------------------------------------------------------------------------------
# Mitsuba helpers
# --------------------------------------------------------------------------------------------------------------------
def vec(*x):
    return np.array(x, dtype=np.float64)


def normalize(v):
    return v / math.sqrt(float(v @ v))


def coordinate_system(n):
    """Mitsuba 3 coordinate_system(n) (Duff et al. 2017) -> (s, t); Frame3f(n) = (s, t, n)"""
    sign = math.copysign(1.0, n[2])
    a = -1.0 / (sign + n[2])
    b = n[0] * n[1] * a
    s = vec(sign * (n[0] * n[0] * a) + 1.0, sign * b, -sign * n[0])
    t = vec(b, n[1] * n[1] * a + sign, -n[1])
    return s, t


class Frame:
    def __init__(self, n, s=None, t=None):
        self.n = n
        if s is None:
            s, t = coordinate_system(n)
        self.s, self.t = s, t

    def to_local(self, v):
        return vec(v @ self.s, v @ self.t, v @ self.n)

    def to_world(self, v):
        return self.s * v[0] + self.t * v[1] + self.n * v[2]


def sh_frame_from_dp_du(n, dp_du):
    """SurfaceInteraction::initialize_sh_frame: s = normalize(dp_du - n (n . dp_du)), t = n x s"""
    s = dp_du - n * float(n @ dp_du)
    if float(s @ s) == 0.0:
        return Frame(n)
    s = normalize(s)
    return Frame(n, s, np.cross(n, s))


def square_to_uniform_disk_concentric(sx, sy):
    x, y = 2.0 * sx - 1.0, 2.0 * sy - 1.0
    if x == 0.0 and y == 0.0:
        return 0.0, 0.0
    q13 = abs(x) < abs(y)
    r, rp = (y, x) if q13 else (x, y)
    phi = 0.25 * math.pi * rp / r
    if q13:
        phi = 0.5 * math.pi - phi
    return r * math.cos(phi), r * math.sin(phi)


def square_to_cosine_hemisphere(sx, sy):
    dx, dy = square_to_uniform_disk_concentric(sx, sy)
    return vec(dx, dy, math.sqrt(max(1.0 - dx * dx - dy * dy, 0.0)))


# --------------------------------------------------------------------------------------------------------------------
# shapes: the closest-hit query scene.ray_intersect (CustomIntegrator.py:309) restated per shape type
# --------------------------------------------------------------------------------------------------------------------
class Parallelogram:
    """p0 + u e1 + v e2, u, v in [0, 1].  A Mitsuba `rectangle` with to_world M is p0 = M(-1,-1,0), e1 = M(2,0,0),
    e2 = M(0,2,0), n = normalize(e1 x e2), dp_du = e1; an OBJ quad (a, b, c, d) is p0 = a, e1 = b - a, e2 = d - a."""

    def __init__(self, p0, e1, e2, bsdf=None, emitter=None, flip=False):
        self.p0, self.e1, self.e2 = (np.asarray(x, dtype=np.float64) for x in (p0, e1, e2))
        self.n = normalize(np.cross(self.e1, self.e2)) * (-1.0 if flip else 1.0)
        self.bsdf, self.emitter = bsdf, emitter
        self.area = math.sqrt(float(np.cross(self.e1, self.e2) @ np.cross(self.e1, self.e2)))

    @classmethod
    def rectangle(cls, M, **kw):
        M = np.asarray(M, dtype=np.float64)
        p = lambda x, y: M[:3, :3] @ vec(x, y, 0.0) + M[:3, 3]
        return cls(p(-1, -1), p(1, -1) - p(-1, -1), p(-1, 1) - p(-1, -1), **kw)

    def intersect(self, o, d):
        """-> (t, u, v, edge_margin) or None"""
        denom = float(self.n @ d)
        if denom == 0.0:
            return None
        t = float(self.n @ (self.p0 - o)) / denom
        if not t > 0.0:
            return None
        q = o + d * t - self.p0
        # (u, v) from the 2 x 2 system in the plane
        a11, a12, a22 = float(self.e1 @ self.e1), float(self.e1 @ self.e2), float(self.e2 @ self.e2)
        b1, b2 = float(q @ self.e1), float(q @ self.e2)
        det = a11 * a22 - a12 * a12
        u, v = (b1 * a22 - b2 * a12) / det, (b2 * a11 - b1 * a12) / det
        if u < 0.0 or v < 0.0 or u > 1.0 or v > 1.0:
            return None
        return t, u, v, min(u, v, 1.0 - u, 1.0 - v)

    def interaction(self, o, d, t, u, v):
        p = self.p0 + self.e1 * u + self.e2 * v
        return p, self.n, self.e1


class Sphere:
    def __init__(self, c, r, bsdf=None, emitter=None):
        self.c, self.r = np.asarray(c, dtype=np.float64), float(r)
        self.bsdf, self.emitter = bsdf, emitter

    def intersect(self, o, d):
        f = o - self.c
        b = float(f @ d)
        disc = b * b - (float(f @ f) - self.r * self.r)
        if disc < 0.0:
            return None
        sq = math.sqrt(disc)
        for t in (-b - sq, -b + sq):
            if t > 0.0:
                return t, 0.0, 0.0, sq / self.r       # margin: how far from grazing
        return None

    def interaction(self, o, d, t, u, v):
        n = normalize(o + d * t - self.c)
        p = self.c + n * self.r
        local = p - self.c
        return p, n, vec(-local[1], local[0], 0.0) * (2.0 * math.pi)     # Sphere: dp_du = 2 pi (-y, x, 0)


class TriMesh:
    """Mitsuba `Mesh` (ply / obj shapes): triangles (p0, p1, p2), geometric normal normalize((p1 - p0) x (p2 - p0)), hit by
    barycentrics (b1, b2); with vertex normals the shading normal is normalize(b0 n0 + b1 n1 + b2 n2)
    (Mesh::compute_surface_interaction).  All triangles are tested at once with NumPy (Moeller-Trumbore in float64)."""

    def __init__(self, vertices, triangles, bsdf=None, vertex_normals=None, emitter=None):
        v = np.asarray(vertices, dtype=np.float64)
        t = np.asarray(triangles, dtype=np.int64)
        self.p0, self.e1, self.e2 = v[t[:, 0]], v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]]
        n = np.cross(self.e1, self.e2)
        self.ng = n / np.linalg.norm(n, axis=1, keepdims=True)
        self.vn = None if vertex_normals is None else np.asarray(vertex_normals, dtype=np.float64)   # [nt, 3, 3]
        self.bsdf, self.emitter = bsdf, emitter
        self._last = -1

    def intersect(self, o, d):
        pvec = np.cross(d, self.e2)
        det = np.einsum("ij,ij->i", self.e1, pvec)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tvec = o - self.p0
            u = np.einsum("ij,ij->i", tvec, pvec) * inv
            qvec = np.cross(tvec, self.e1)
            v = (qvec @ d) * inv
            t = np.einsum("ij,ij->i", self.e2, qvec) * inv
            ok = (det != 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0)
        if not ok.any():
            return None
        tt = np.where(ok, t, np.inf)
        k = int(np.argmin(tt))
        self._last = k
        # margin: distance of the barycentrics from the triangle's edges, and of the runner-up hit in t (shared edges)
        others = np.delete(tt, k)
        gap = (others.min() - tt[k]) / max(tt[k], 1e-30) if len(others) and np.isfinite(others.min()) else math.inf
        return float(tt[k]), float(u[k]), float(v[k]), min(float(u[k]), float(v[k]), float(1.0 - u[k] - v[k]), 1e3 * gap + (1.0 if gap > 1e-6 else 0.0))

    def interaction(self, o, d, t, u, v):
        k = self._last
        p = self.p0[k] + self.e1[k] * u + self.e2[k] * v
        return p, self.ng[k], self.e1[k]

    def shading_normal(self, u, v):
        k = self._last
        if self.vn is None:
            return self.ng[k]
        m = self.vn[k, 0] * (1.0 - u - v) + self.vn[k, 1] * u + self.vn[k, 2] * v
        return normalize(m)


def ray_intersect(shapes, o, d, tmax=math.inf):
    best = None
    for s in shapes:
        h = s.intersect(o, d)
        if h is not None and h[0] <= tmax and (best is None or h[0] < best[1][0]):
            best = (s, h)
    return best


def spawn_origin(p, n, d):
    """SurfaceInteraction::offset_p(d): p + n * copysign((1 + max|p|) * RayEpsilon, n . d)"""
    mag = (1.0 + float(np.max(np.abs(p)))) * RAY_EPSILON
    return p + n * math.copysign(mag, float(n @ d))


# --------------------------------------------------------------------------------------------------------------------
# UltraBSDF (CustomBSDF.py) -- literal arithmetic, quirks A1-A9 of SURVEY.md App. A included
# --------------------------------------------------------------------------------------------------------------------
MEDIUM_Z = 1.2   # CustomBSDF.py:105


def ultra_ggx_sample(roughness, wi_world, n_world, sample):
    frame = Frame(n_world)                                                    # :32
    wi = frame.to_local(wi_world)                                             # :33
    alpha = roughness                                                         # :34
    wi_stretched = normalize(vec(alpha * wi[0], alpha * wi[1], wi[2]))        # :37-38
    inv_len = 1.0 / math.sqrt(max(1.0 - wi_stretched[2] * wi_stretched[2], 1e-7))   # :41
    T1 = vec(wi_stretched[1] * inv_len, -wi_stretched[0] * inv_len, 0.0)      # :42-44
    T2 = np.cross(wi_stretched, T1)                                           # :45
    dx, dy = square_to_uniform_disk_concentric(sample, sample)                # :48  scalar Float -> Point2f(s, s)
    S = 0.5 * (1.0 + wi_stretched[2])                                         # :51
    dy = (1.0 - S) * math.sqrt(max(1.0 - dx * dx, 0.0)) + S * dy              # :52
    m_stretched = dx * T1 + dy * T2 + math.sqrt(max(1.0 - dx * dx - dy * dy, 0.0)) * wi_stretched   # :55
    m = normalize(vec(alpha * m_stretched[0], alpha * m_stretched[1], m_stretched[2]))             # :56-59
    ultra_ggx_sample.cancellation = 1.0 - wi_stretched[2] * wi_stretched[2]   # (not reference code) conditioning of :41
    return m


def ultra_bsdf_sample(impedance, roughness, si_wi, si_n, si_sh_n, sh_frame, sample1, sample2):
    """CustomBSDF.py:87-175 -> dict(wo, pdf, a_resp, chosen, reflect, margin)"""
    incident_direction = si_wi                                                # :90
    surface_normal = si_sh_n                                                  # :91
    m = ultra_ggx_sample(roughness, si_wi, si_n, sample1)                     # :95
    flip_margin = abs(float(m @ incident_direction))
    if not (float(m @ incident_direction) < 0.0):                             # :100
        m = -m
    cos_wi_m = float(incident_direction @ m)                                  # :101
    entering = float(m @ incident_direction) > 0.0                            # :104 (never true after :100)
    Z1 = MEDIUM_Z if entering else impedance                                  # :106
    Z2 = impedance if entering else MEDIUM_Z                                  # :107
    snells_ratio = Z1 / Z2                                                    # :111
    cosTr = abs(float(m @ incident_direction))                                # :119
    sqrt_arg = 1.0 - (snells_ratio ** 2) * (1.0 - cosTr ** 2)                 # :120
    cosTt = math.sqrt(max(sqrt_arg, 0.0))                                     # :121
    denom = Z1 * cosTr + Z2 * cosTt                                           # :122
    Ar = (Z1 * cosTr - Z2 * cosTt) / denom                                    # :123
    At = 1.0 - Ar                                                             # :124
    reflected_direction = incident_direction + 2.0 * cos_wi_m * m             # :130
    transmission_direction = snells_ratio * reflected_direction + (snells_ratio * cosTr - cosTt) * m   # :131
    tir = sqrt_arg < 0.0                                                      # :137
    prob_reflect = Ar * Ar                                                    # :142
    prob_reflect_bool = sample2 < prob_reflect                                # :144
    select_reflect = True if tir else prob_reflect_bool                       # :145
    chosen_dir = reflected_direction if select_reflect else transmission_direction   # :147
    pdf_m = 1.0                                                               # :153 ggx_pdf -> pdf_max / pdf_max (:81-82)
    pdf_reflect = pdf_m / (4.0 * abs(cos_wi_m))                               # :154
    cos_wo_m = float(transmission_direction @ m)                              # :155
    abs_n_wi = abs(float(surface_normal @ incident_direction))                # :156
    abs_n_wo = max(abs(float(surface_normal @ transmission_direction)), 1e-7)  # :157
    pdf_trans = pdf_m * snells_ratio ** 2 * abs(cos_wo_m) / (abs_n_wi * abs_n_wo)   # :158
    wo = sh_frame.to_local(chosen_dir)                                        # :165 si.to_local
    pdf = pdf_reflect if select_reflect else pdf_trans                        # :166
    acoustic_response_amp = Ar if select_reflect else At                      # :170
    # how far the sample is from a discontinuity / a clamped denominator (f32 and f64 may then take different sides)
    # (:41 subtracts two numbers that agree to 1 - z^2: near normal incidence f32 keeps few digits of it)
    margin = min(flip_margin, abs(sqrt_arg), math.inf if tir else abs(sample2 - prob_reflect), 10.0 * ultra_ggx_sample.cancellation,
                 math.inf if select_reflect else min(abs_n_wi, abs(float(surface_normal @ transmission_direction))))
    return dict(wo=wo, pdf=pdf, a_resp=acoustic_response_amp, chosen=chosen_dir, reflect=bool(select_reflect), tir=bool(tir),
                Ar=Ar, margin=margin)


# --------------------------------------------------------------------------------------------------------------------
# UltraIntegrator._trace_single_ray (CustomIntegrator.py:262-376)
# --------------------------------------------------------------------------------------------------------------------
def us_setup(P):
    """:237-257 -> elem_x (float32, as :248 builds it), angles_rad, tx_delay (float32, :257)"""
    n = int(P["n_elements"])
    angles_rad = np.deg2rad(np.asarray(P["angles_deg"], dtype=np.float64))                          # :247
    elem_x = np.float32(P["pitch"]) * (np.arange(n, dtype=np.float32) - np.float32((n - 1) / 2))     # :248
    ang_grid, elem_grid = np.meshgrid(angles_rad, elem_x.astype(np.float64), indexing="ij")         # :251
    tx_delay = (elem_grid * np.sin(ang_grid)) / float(P["sound_speed"])                             # :254
    return elem_x.astype(np.float64), angles_rad, tx_delay.astype(np.float32).astype(np.float64)    # :257


def us_trace_single_ray(shapes, sensor_T, P, angle_idx, elem_idx, draws, variant="scalar", primary=None):
    """draws(depth) -> (u_recv, s1, s2, u_rr): the four uniforms of one bounce (:319, :337, :365).
    primary: None = the integrator's own ray (:264-273); else (o, d, ray time, weight) of CustomEmitter.sample_ray in the
    transducer's frame (emitter_sample_ray below): the path starts there -- origin and direction taken to the world like :273,
    the ray's time as its initial time of flight (t0 = 0), its weight as a factor of every echo the path deposits -- Mitsuba's
    SamplingIntegrator::render_sample multiplies what Integrator.sample returns by the weight of the sampled ray, the path's own
    amplitude still starts at 1 (:276) -- (the library's PBRT_US_PRIMARY_EMITTER; the reference itself never connects its emitter
    to its integrator).
    variant "scalar": _trace_single_ray of simulate_acquisition_parallel (:262-376, what USMain.py calls);
    variant "drjit": the body of simulate_acquisition's dr.while_loop (:137-226), where it differs -- the draws are taken
    while the loop body is TRACED (llvm_ad_mono, USMain.py:12), i.e. once per ray: every bounce sees draws(0) (:153,173-174,
    219); tof is never updated, the echo time uses tof + distance / c of the current segment only (:165); the time bin is
    clamped to [0, T - 1] instead of dropped (:191-193); roulette without the abs, strict <, atten zeroed on death (:219-224).
    -> list of per-bounce records."""
    n_angles, n_elements = len(P["angles_deg"]), int(P["n_elements"])
    fs_scalar, c_scalar, time_samples_scalar = float(P["fs"]), float(P["sound_speed"]), int(P["time_samples"])
    num_rays = n_angles * n_elements                                                                # :243
    elem_x, angles_rad, tx_delay = us_setup(P)
    R, tr = np.asarray(sensor_T, dtype=np.float64)[:3, :3], np.asarray(sensor_T, dtype=np.float64)[:3, 3]
    a_rad = float(angles_rad[angle_idx])                                                            # :265
    x_elem = float(elem_x[elem_idx])                                                                # :266
    t0 = float(tx_delay[angle_idx, elem_idx])                                                       # :267
    origin = vec(x_elem, 0.0, 0.0)                                                                  # :270
    direction = vec(math.sin(a_rad), 0.0, math.cos(a_rad))                                          # :271
    ray_o, ray_d = R @ origin + tr, normalize(R @ direction)                                        # :273
    amp, atten, tof, geo_len, depth, active = 1.0, 1.0, 0.0, 0.0, 0, True                           # :276-281
    ray_weight = 1.0
    if primary is not None:
        e_o, e_d, e_time, e_weight = primary                                                        # CustomEmmitter.py:100-107
        ray_o, ray_d = R @ np.asarray(e_o, dtype=np.float64) + tr, normalize(R @ np.asarray(e_d, dtype=np.float64))
        tof, t0, ray_weight = float(e_time), 0.0, float(e_weight)
    trans_normal_world = normalize(R @ vec(0.0, 0.0, 1.0))                                          # :292, :369
    alpha_m, alpha_c = math.radians(P["main_beam_angle"]), math.radians(P["cutoff_angle"])          # :345
    out = []
    while active and depth < P["max_depth"] and geo_len < 0.2:                                      # :307
        hit = ray_intersect(shapes, ray_o, ray_d)                                                   # :309
        if hit is None:                                                                             # :311-312
            break
        shape, (t, u, v, edge_margin) = hit
        p, n, dp_du = shape.interaction(ray_o, ray_d, t, u, v)
        # si.sh_frame.n: the geometric normal, except on a Mesh with vertex normals (Mesh::compute_surface_interaction interpolates them)
        n_sh = shape.shading_normal(u, v) if isinstance(shape, TriMesh) else n
        sh = sh_frame_from_dp_du(n_sh, dp_du)                   # si.sh_frame
        si_wi = sh.to_local(-ray_d)                             # SurfaceInteraction::finalize: wi = to_local(-ray.d)
        drjit = variant == "drjit"
        distance = t                                                                                # :314
        geo_len += distance                                                                         # :315
        if not drjit:
            tof += distance / c_scalar                                                              # :316
        u_recv, s1, s2, u_rr = draws(0 if drjit else depth)
        recv_idx = min(int(u_recv * n_elements), n_elements - 1)                                    # :319 rng.integers
        target_w = R @ vec(float(elem_x[recv_idx]), 0.0, 0.0) + tr                                  # :320-321
        sec_dir = normalize(target_w - p)                                                           # :322
        vis = ray_intersect(shapes, spawn_origin(p, n, sec_dir), sec_dir)                           # :324 (unbounded, B7)
        visible = vis is None                                                                       # :325
        atten *= math.exp(-P["attenuation"] * P["frequency"] * 1e-6 * distance / 8.686)             # :328
        tof_to_intersection = tof + distance / c_scalar if drjit else tof                           # drjit :165
        total_time = t0 + tof_to_intersection + math.sqrt(float((target_w - p) @ (target_w - p))) / c_scalar   # :329
        phase = 2.0 * math.pi * P["frequency"] * total_time                                         # :330
        bs = ultra_bsdf_sample(shape.bsdf["impedance"], shape.bsdf["roughness"], si_wi, n, n_sh, sh, s1, s2)   # :338
        cos_theta = float(n_sh @ -ray_d)                                                            # :340 (si.sh_frame.n)
        amp *= bs["a_resp"] * cos_theta * max(bs["pdf"], 1e-6)                                      # :341
        # :345  directivity_weight_i(sec_dir, ...) * directivity_weight_o(ray.d, si.sh_frame.n, num_rays)
        alpha = abs(math.acos(max(-1.0, min(1.0, float(trans_normal_world @ -sec_dir)))))           # :293-295
        mid_cond = (alpha_c - alpha) / (alpha_c - alpha_m)                                          # :297
        w_i = 1.0 if alpha <= alpha_m else (mid_cond if alpha <= alpha_c else 0.0)                  # :299-302
        w_o = float(ray_d @ n_sh) / num_rays                                                        # :286-287 (si.sh_frame.n, :345)
        fd = w_i * w_o
        envelope = atten * amp * fd * ray_weight      # (ray_weight: 1, or the emitter ray's -- applied to what the loop deposits)
        pressure_scalar = envelope * math.sin(phase)                                                # :348
        t_float = total_time * fs_scalar                                                            # :351
        t_idx = int(round(t_float))                                                                 # :352 (half to even)
        if drjit:
            t_idx = min(max(t_idx, 0), time_samples_scalar - 1)                                     # drjit :191-193
        deposited = 0 <= t_idx < time_samples_scalar and visible                                    # :353
        new_dir = sh.to_world(bs["wo"])                                                             # :358
        nd = normalize(new_dir)
        ray_o, ray_d_next = spawn_origin(p, n, nd), nd                                              # :359
        depth += 1                                                                                  # :361
        atten_before_rr = atten
        if drjit:
            rr_prob = min(atten * amp, 1.0)                                                         # drjit :220
            survive = u_rr < rr_prob                                                                # :221
            atten = atten / rr_prob if survive else 0.0                                             # :224
        else:
            rr_prob = min(abs(atten * amp), 1.0)                                                    # :364
            survive = True                                                                          # (B5 repaired)
            if u_rr > rr_prob:                                                                      # :365-366
                survive = False
            atten /= rr_prob                                                                        # :367
        cos_min = math.cos(math.radians(P["cutoff_angle"]))                                         # :370
        within_angle = float(ray_d_next @ trans_normal_world) >= cos_min                            # :371
        path_ok = geo_len < 0.2                                                                     # :372
        depth_ok = depth < P["max_depth"]                                                           # :373
        active = active and within_angle and path_ok and depth_ok and survive                       # :376
        margin = min(edge_margin, bs["margin"], abs(t_float - math.floor(t_float) - 0.5), abs(u_rr - rr_prob),
                     abs(float(ray_d_next @ trans_normal_world) - cos_min), abs(alpha - alpha_m), abs(alpha - alpha_c),
                     abs(u_recv * n_elements - round(u_recv * n_elements)) + (0.0 if 0 < round(u_recv * n_elements) < n_elements else 1.0))
        out.append(dict(depth=depth - 1, recv=recv_idx, t_idx=t_idx, visible=bool(visible), deposited=bool(deposited),
                        pressure=pressure_scalar, envelope=envelope, pdf=bs["pdf"], a_resp=bs["a_resp"], reflect=bs["reflect"],
                        tir=bs["tir"], wi=si_wi.tolist(), n=n.tolist(), sh_n=n_sh.tolist(), wo=bs["wo"].tolist(), new_dir=nd.tolist(), t=t,
                        p=p.tolist(), amp=amp, atten=atten_before_rr, rr_prob=rr_prob, survive=bool(survive),
                        within=bool(within_angle), active=bool(active), s1=s1, s2=s2, margin=margin,
                        sh_s=sh.s.tolist(), sh_t=sh.t.tolist(), shape=[i for i, s_ in enumerate(shapes) if s_ is shape][0]))
        ray_d = ray_d_next
    return out


# --------------------------------------------------------------------------------------------------------------------
# radiance mode: Mitsuba `path`
# --------------------------------------------------------------------------------------------------------------------
def mis_weight(pdf_a, pdf_b):
    """power heuristic a^2 / (a^2 + b^2), 0 when not finite (Mitsuba path.cpp mis_weight)"""
    a2 = pdf_a * pdf_a
    w = a2 / (a2 + pdf_b * pdf_b) if (a2 + pdf_b * pdf_b) > 0.0 else 0.0
    return w if math.isfinite(w) else 0.0


def fresnel_dielectric(cos_theta_i, eta):
    """Mitsuba fresnel(cos_theta_i, eta) -> (r, cos_theta_t, eta_it, eta_ti)"""
    outside = cos_theta_i >= 0.0
    rcp_eta = 1.0 / eta
    eta_it, eta_ti = (eta, rcp_eta) if outside else (rcp_eta, eta)
    cos_theta_t_sqr = 1.0 - (1.0 - cos_theta_i * cos_theta_i) * eta_ti * eta_ti
    cia, cta = abs(cos_theta_i), math.sqrt(max(cos_theta_t_sqr, 0.0))
    a_s = (cia - eta_it * cta) / (cia + eta_it * cta)
    a_p = (cta - eta_it * cia) / (cta + eta_it * cia)
    r = 0.5 * (a_s * a_s + a_p * a_p)
    if eta == 1.0:
        r = 0.0
    elif cia == 0.0:
        r = 1.0
    return r, math.copysign(cta, -cos_theta_i), eta_it, eta_ti


def perspective_ray(cam, sx, sy):
    """Mitsuba `perspective`::sample_ray for a film position (sx, sy) in [0,1)^2 (App. D)"""
    tx = math.tan(math.radians(cam["x_fov"]) / 2.0)
    ty = tx * cam["height"] / cam["width"]
    d_cam = normalize(vec((1.0 - 2.0 * sx) * tx, (1.0 - 2.0 * sy) * ty, 1.0))
    M = np.asarray(cam["to_world"], dtype=np.float64)
    d = normalize(M[:3, :3] @ d_cam)
    inv_z = 1.0 / d_cam[2]
    return M[:3, 3] + d * (cam["near"] * inv_z), d, (cam["far"] - cam["near"]) * inv_z


def look_at(origin, target, up):
    """Mitsuba Transform::look_at: columns (left, new_up, dir, origin)"""
    o, t, u = (np.asarray(x, dtype=np.float64) for x in (origin, target, up))
    d = normalize(t - o)
    left = normalize(np.cross(u, d))
    new_up = np.cross(d, left)
    M = np.eye(4)
    M[:3, 0], M[:3, 1], M[:3, 2], M[:3, 3] = left, new_up, d, o
    return M


class PointLight:
    """Mitsuba `point` emitter: delta position, radiant intensity I -> incident radiance I / dist^2"""

    def __init__(self, position, intensity):
        self.position, self.intensity = np.asarray(position, dtype=np.float64), np.asarray(intensity, dtype=np.float64)


def path_radiance(shapes, lights, o, d, tmax, key, seed, max_depth, rr_depth):
    """Mitsuba path.cpp sample(): emission with MIS, emitter sampling with a shadow ray, BSDF sampling, roulette.
    lights: the emitting Parallelograms (one `area` emitter each).  key = (a, b): draws come from rng4(a, b, block, seed)
    with block 1 + 2k = emitter sample of bounce k (emitter pick, primitive pick, 2-D point) and 2 + 2k = BSDF sample
    (lobe, 2-D direction) + roulette -- the build's keying.  -> (rgb, margin)"""
    throughput, result = np.ones(3), np.zeros(3)
    eta, depth = 1.0, 0
    prev_bsdf_pdf, prev_bsdf_delta = 1.0, True
    nE = len(lights)
    margin = math.inf
    while True:
        hit = ray_intersect(shapes, o, d, tmax)
        if hit is None:
            break
        shape, (t, u, v, edge_margin) = hit
        margin = min(margin, edge_margin)
        p, n, _ = shape.interaction(o, d, t, u, v)
        ns = shape.shading_normal(u, v) if hasattr(shape, "shading_normal") else n      # si.sh_frame.n
        # ---- emitter hit: result += throughput * Le * mis   (area emitters are one-sided: eval = radiance if n . wi > 0)
        if shape.emitter is not None:
            cos_l = -float(ns @ d)
            if cos_l > 0.0:
                w = 1.0
                if not prev_bsdf_delta:
                    # pdf of having sampled this point through emitter sampling: dist^2 / (cos * area) / #emitters
                    pdf_em = (t * t) / (cos_l * shape.area) / nE
                    w = mis_weight(prev_bsdf_pdf, pdf_em)
                result = result + throughput * np.asarray(shape.emitter["radiance"]) * w
        if depth + 1 >= max_depth:
            break
        frame = Frame(ns)                                          # D13: coordinate_system(sh_frame.n)
        wi = frame.to_local(-d)
        bsdf = shape.bsdf
        # ---- emitter sampling (only BSDFs with a smooth component)
        L = None
        if bsdf["type"] == "diffuse" and nE > 0:
            ue = rng4(key[0], key[1], 1 + 2 * depth, seed)
            ei = min(int(ue[0] * nE), nE - 1)
            L = lights[ei]
            if isinstance(L, PointLight):
                dv = L.position - p
                dist2 = float(dv @ dv)
                dist = math.sqrt(dist2)
                dl = dv / dist
                wo = frame.to_local(dl)
                if wi[2] > 0.0 and wo[2] > 0.0:
                    f_cos = np.asarray(bsdf["reflectance"]) * (wo[2] / math.pi)
                    so = spawn_origin(p, n, dl)
                    sv = L.position - so
                    sd = math.sqrt(float(sv @ sv))
                    occ = ray_intersect(shapes, so, sv / sd, sd * (1.0 - SHADOW_EPSILON))
                    if occ is None:   # delta emitter: no MIS; weight = I / dist^2 / (selection probability 1 / nE)
                        result = result + throughput * f_cos * (L.intensity / dist2) * nE
                    margin = min(margin, 1.0 if occ is None else occ[1][3])
                L = None
        if L is not None and bsdf["type"] == "diffuse" and nE > 0:
            q = L.p0 + L.e1 * ue[2] + L.e2 * ue[3]                 # D10: one uniform point on the parallelogram
            dv = q - p
            dist2 = float(dv @ dv)
            dist = math.sqrt(dist2)
            dl = dv / dist
            cos_l = -float(L.n @ dl)
            if cos_l > 0.0:
                ds_pdf = dist2 / (cos_l * L.area) / nE             # solid-angle density x emitter selection
                wo = frame.to_local(dl)
                if wi[2] > 0.0 and wo[2] > 0.0:
                    f_cos = np.asarray(bsdf["reflectance"]) * (wo[2] / math.pi)
                    bsdf_pdf = wo[2] / math.pi
                    so = spawn_origin(p, n, dl)                    # Interaction::spawn_ray_to(q)
                    sv = q - so
                    sd = math.sqrt(float(sv @ sv))
                    occ = ray_intersect([s for s in shapes if s is not L], so, sv / sd, sd * (1.0 - SHADOW_EPSILON))
                    if occ is None:
                        result = result + throughput * f_cos * (np.asarray(L.emitter["radiance"]) / ds_pdf) * mis_weight(ds_pdf, bsdf_pdf)
                    margin = min(margin, shadow_clearance(shapes, L, so, sv / sd, sd))
        # ---- BSDF sampling
        ub = rng4(key[0], key[1], 2 + 2 * depth, seed)
        if bsdf["type"] == "diffuse":
            if not wi[2] > 0.0:
                break
            wo = square_to_cosine_hemisphere(ub[1], ub[2])
            pdf = wo[2] / math.pi
            if not pdf > 0.0:
                break
            weight, delta, bs_eta = np.asarray(bsdf["reflectance"], dtype=np.float64), False, 1.0
        elif bsdf["type"] == "conductor":                          # no parameters: perfect mirror
            if not wi[2] > 0.0:
                break
            wo, pdf, weight, delta, bs_eta = vec(-wi[0], -wi[1], wi[2]), 1.0, np.ones(3), True, 1.0
        else:                                                      # dielectric
            r, cos_t, eta_it, eta_ti = fresnel_dielectric(wi[2], bsdf["eta"])
            margin = min(margin, abs(ub[0] - r))
            delta = True
            if ub[0] <= r:
                wo, pdf, weight, bs_eta = vec(-wi[0], -wi[1], wi[2]), r, np.ones(3), 1.0
            else:
                wo, pdf, weight, bs_eta = vec(-eta_ti * wi[0], -eta_ti * wi[1], cos_t), 1.0 - r, np.ones(3) * eta_ti ** 2, eta_it
        throughput = throughput * weight
        eta *= bs_eta
        d = frame.to_world(wo)
        o = spawn_origin(p, n, d)
        tmax = math.inf
        prev_bsdf_pdf, prev_bsdf_delta = pdf, delta
        depth += 1
        tm = float(np.max(throughput))
        if depth >= rr_depth:
            qrr = min(tm * eta * eta, 0.95)
            margin = min(margin, abs(ub[3] - qrr))
            throughput = throughput / qrr
            if not ub[3] < qrr:
                break
        if tm == 0.0:
            break
    return result, margin


def shadow_clearance(shapes, light, o, d, dist):
    """how far the shadow segment stays from changing its visibility: smallest edge margin of the occluders' hits /
    misses along the segment (coarse: spheres by their closest approach, quads by their in-plane coordinates)"""
    m = math.inf
    for s in shapes:
        if s is light:
            continue
        if isinstance(s, Sphere):
            f = o - s.c
            tca = -float(f @ d)
            if 0.0 < tca < dist:
                d2 = float(f @ f) - tca * tca
                m = min(m, abs(math.sqrt(max(d2, 0.0)) - s.r) / s.r)
        else:
            denom = float(s.n @ d)
            if denom != 0.0:
                t = float(s.n @ (s.p0 - o)) / denom
                if 0.0 < t < dist * 1.001:
                    q = o + d * t - s.p0
                    a11, a12, a22 = float(s.e1 @ s.e1), float(s.e1 @ s.e2), float(s.e2 @ s.e2)
                    b1, b2 = float(q @ s.e1), float(q @ s.e2)
                    det = a11 * a22 - a12 * a12
                    u, v = (b1 * a22 - b2 * a12) / det, (b2 * a11 - b1 * a12) / det
                    m = min(m, abs(u), abs(v), abs(1.0 - u), abs(1.0 - v), abs(t - dist) / dist + (0.0 if 0 <= u <= 1 and 0 <= v <= 1 else 1.0))
    return m


# --------------------------------------------------------------------------------------------------------------------
# SURVEY rows a13 / a14 -- the API-only leaf classes.  CustomEmitter from the reference's SOURCE (CustomEmmitter.py:30-107, line by
# line); UltraSensor from SURVEY App. C (its source survives only as CPython-3.12 bytecode, read statically).  float64, the random
# numbers are INPUTS.  dr.linspace(a, b, n)[i] = a + i (b - a) / (n - 1); mi.warp.square_to_uniform_hemisphere(s) =
# (dx k, dy k, z) with (dx, dy) = concentric disk, z = 1 - |disk|^2, k = sqrt(z + 1)  (Mitsuba warp.h).
# --------------------------------------------------------------------------------------------------------------------
def emitter_element_geometry(P, i):
    """CustomEmmitter.py:30-49 compute_element_geometry, element i -> (position, unit normal)"""
    N = P["number_of_elements"]
    if P["radius"] == 0.0:                                                     # :33
        a, b = -(N - 1) / 2 * P["pitch"], (N - 1) / 2 * P["pitch"]              # :34-36
        x = a + i * (b - a) / (N - 1) if N > 1 else a
        pos, nrm = vec(x, 0.0, 0.0), vec(0.0, 0.0, 1.0)                         # :37-38
    else:
        span = math.radians(P["opening_angle"])                                # :42
        th = -span / 2 + i * span / (N - 1) if N > 1 else -span / 2            # :43
        pos = vec(P["radius"] * math.sin(th), 0.0, P["radius"] * math.cos(th))  # :44-46
        nrm = vec(math.sin(th), 0.0, math.cos(th))                             # :47
    return pos, normalize(nrm)                                                 # :49


def emitter_sample_position(P, sample1, sample2):
    """CustomEmmitter.py:51-79 -> (p, n, pdf)"""
    N = P["number_of_elements"]
    idx = int(min(math.floor(sample1 * N), N - 1))                             # :56-57
    center, normal = emitter_element_geometry(P, idx)                          # :60-61
    dx = (sample2[0] - 0.5) * P["element_width"]                               # :64
    dy = (sample2[1] - 0.5) * P["element_height"]                              # :65
    p = center + vec(dx, dy, 0.0)                                              # :68
    pdf = 1.0 / (N * P["element_width"] * P["element_height"])                 # :77
    return p, normal, pdf


def emitter_sample_ray(P, time, sample1, sample2, sample3):
    """CustomEmmitter.py:81-107 -> (o, d, ray time, weight, pdf_pos)"""
    p, n, pdf = emitter_sample_position(P, sample1, sample2)                   # :82
    psi_min, psi_max = math.radians(P["steering_angle_min"]), math.radians(P["steering_angle_max"])   # :85-86
    psi = psi_min + sample3 * (psi_max - psi_min)                              # :87
    d = vec(math.sin(psi), 0.0, math.cos(psi))                                 # :90
    delta_t = time + (-(p[0] * math.sin(psi)) / P["speed_of_sound"])           # :93-94
    fd = max(0.0, float(d @ n))                                                # :97
    weight = fd / (P["number_of_elements"] * P["number_of_rays_per_element"])   # :17, :98
    return p, d, delta_t, weight, pdf


def square_to_uniform_hemisphere(sx, sy):
    dx, dy = square_to_uniform_disk_concentric(sx, sy)
    z = 1.0 - (dx * dx + dy * dy)
    k = math.sqrt(z + 1.0)
    return vec(dx * k, dy * k, z)


def ultra_sensor_sample_ray(P, T, time, wavelength_sample, position_sample, aperture_sample, use_hemisphere_warp=True):
    """SURVEY App. C, UltraSensor.sample_ray (bytecode src lines 37-90) -> (o, d, weight).  T: 4x4 to_world."""
    N = P["num_elements_lateral"]
    idx = min(math.floor(position_sample[0] * N), N - 1)
    if math.isinf(P["radius"]):
        ex, ez = -((N - 1) * P["pitch"]) / 2 + idx * P["pitch"], 0.0
    else:
        th = (idx - N / 2) * (P["pitch"] / P["radius"])
        ex, ez = P["radius"] * math.sin(th), P["radius"] * (1.0 - math.cos(th))
    ox = (aperture_sample[0] - 0.5) * P["element_width"]
    oy = (aperture_sample[1] - 0.5) * P["element_height"]
    o_local = vec(ex + ox, oy, ez)
    if use_hemisphere_warp:
        d_local = square_to_uniform_hemisphere(aperture_sample[0], aperture_sample[1])
    else:
        phi, ct = 2.0 * math.pi * position_sample[1], wavelength_sample
        st = math.sqrt(max(0.0, 1.0 - ct * ct))
        d_local = vec(st * math.cos(phi), st * math.sin(phi), ct)
    o_world = T[:3, :3] @ o_local + T[:3, 3]
    d_world = normalize(T[:3, :3] @ d_local)
    weight = math.cos(2.0 * math.pi * P["center_frequency"] * time) * (abs(d_local[2]) * P["directivity"])
    return o_world, d_world, weight
