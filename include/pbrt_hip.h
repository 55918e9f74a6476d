/*
 * pbrt_hip.h -- C-ABI of libpbrt_hip.so, the MI355X (gfx950) Monte-Carlo ray-transport engine.
 *
 * This is the drop-in boundary of the hot path.  The reference
 * (ReaganCardoza/Physics-Based-Ray-Tracing) has NO FFI: its boundary is the Mitsuba-3 Python
 * plugin API.  Each entry point below therefore cites the reference *Python* interface it
 * replaces (file:line relative to the reference root); the Python mirror of that interface
 * lives in physics-based-ray-tracing_amd/ and reaches this library through ctypes only.
 *
 * Conventions
 *  - extern "C", plain pointers + sizes, no C++/torch types.
 *  - every function returns int: 0 = OK, <0 = error class (PBRT_E_*); the message is available
 *    from pbrt_last_error().  Nothing throws or exits across the ABI.
 *  - batched leaf ops take HOST pointers to C-contiguous SoA arrays ([component][n], f32/u32);
 *    the library stages them through its own device buffers.
 *  - *_dev variants take DEVICE pointers (e.g. torch tensor data_ptr(), or pbrt_dev_alloc) and keep data in HBM.
 *  - one pbrt_ctx per device; calls on a ctx are not re-entrant; calls are synchronous on return (the ctx
 *    stream has been synchronised) EXCEPT the image-formation *_dev entry points of ABI 5
 *    (pbrt_das_beamform_dev, pbrt_envelope_dev, pbrt_log_compress_dev, pbrt_us_apply_pulse_dev), pbrt_us_acquire_queue_dev,
 *    pbrt_scene_update_material and pbrt_dev_upload: those queue their work on the ctx stream, in call order behind everything queued before,
 *    and return; pbrt_ctx_synchronize / pbrt_dev_download wait for it.  A caller that hands in memory of
 *    another runtime's stream (a torch tensor) orders the two streams itself.
 *  - the caller owns every buffer it passes; the library owns device memory behind the opaque
 *    handles and retains no caller pointer past a call.
 */
#ifndef PBRT_HIP_H
#define PBRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBRT_ABI_VERSION 5

/* ---- error classes ------------------------------------------------------------------------ */
#define PBRT_OK 0
#define PBRT_E_INVALID (-1) /* bad argument / inconsistent descriptor            */
#define PBRT_E_DEVICE (-2)  /* HIP runtime error (no device, launch failure ...) */
#define PBRT_E_NOMEM (-3)   /* host or device allocation failed                  */
#define PBRT_E_UNSUPPORTED (-4)

/* ---- primitives --------------------------------------------------------------------------- */
/* One 64-byte record per primitive.  World space, f32.
 *   TRIANGLE       g = v0[3], e1 = v1-v0 [3], e2 = v2-v0 [3], n = normalize(e1 x e2) [3]
 *   SPHERE         g = centre[3], radius, 8 x 0
 *   PARALLELOGRAM  g = corner[3], e1[3], e2[3], n[3]   (Mitsuba 'rectangle' under to_world:
 *                  corner = T(-1,-1,0), e1 = T(1,-1,0)-corner, e2 = T(-1,1,0)-corner)
 *   CONE           g = row-major 3x4 world -> object matrix of the closed unit cone (apex (0,0,1), base disc of
 *                  radius 1 in z = 0): any affine to_world, e.g. the 0.06 / 0.06 / 0.10 scale of
 *                  MitsubaScenes/Cone_Box.xml:36-47.  Hits report u = 0 (lateral surface) / 1 (base disc), v = 0.
 * The scenes that feed these: scenes/cbox.xml (12 triangles + 2 spheres), scenes/simple.xml
 * (teapot.ply, 2256 triangles), MitsubaScenes/ *.xml (sphere / rectangle / cone),
 * TestRing/TestRing.obj (1152 triangles).
 */
#define PBRT_PRIM_TRIANGLE 0u
#define PBRT_PRIM_SPHERE 1u
#define PBRT_PRIM_PARALLELOGRAM 2u
#define PBRT_PRIM_CONE 3u /* analytic; cannot carry an area emitter (like SPHERE) */

typedef struct pbrt_prim {
    float g[12];
    uint32_t type;     /* PBRT_PRIM_*                                   */
    uint32_t material; /* index into pbrt_scene_desc.materials          */
    int32_t emitter;   /* index into pbrt_scene_desc.emitters, -1: none */
    uint32_t shape;    /* caller's shape id (returned untouched)        */
} pbrt_prim;

/* ---- materials (BSDFs) -------------------------------------------------------------------- */
/* DIFFUSE     p[0..2] = reflectance                      (Mitsuba 'diffuse', scenes/cbox.xml:36-50)
 * CONDUCTOR   p[0..2] = specular_reflectance, perfect mirror (Mitsuba 'conductor' default,
 *                                                          scenes/cbox.xml:54)
 * DIELECTRIC  p[0] = eta = int_ior / ext_ior             (Mitsuba 'dielectric', scenes/cbox.xml:52)
 * ULTRA       p[0] = impedance, p[1] = roughness, p[2] = medium_z (1.2)
 *                                                        (UltraBSDF, CustomBSDF.py:7-26,105)
 * NONE        absorbs everything
 */
#define PBRT_MAT_DIFFUSE 0u
#define PBRT_MAT_CONDUCTOR 1u
#define PBRT_MAT_DIELECTRIC 2u
#define PBRT_MAT_ULTRA 3u
#define PBRT_MAT_NONE 4u

typedef struct pbrt_material {
    uint32_t type;
    float p[7];
} pbrt_material;

/* ---- emitters ----------------------------------------------------------------------------- */
/* AREA   radiance; its primitives are light_prims[first .. first+count) with the inclusive,
 *        normalised area CDF light_cdf[first .. first+count); area = summed surface area.
 * POINT  radiance = intensity, pos = position.
 */
#define PBRT_EMIT_AREA 0u
#define PBRT_EMIT_POINT 1u

typedef struct pbrt_emitter {
    uint32_t type;
    float radiance[3];
    float pos[3];
    uint32_t first;
    uint32_t count;
    float area;
    float pad[2];
} pbrt_emitter;

#define PBRT_ACCEL_AUTO 0u  /* brute force when n_prims <= 32, LDS-resident BVH otherwise */
#define PBRT_ACCEL_BRUTE 1u /* uniform loop over all primitives (scalar loads)            */
#define PBRT_ACCEL_BVH 2u   /* BVH4 (child boxes on an 8-bit grid, 40-byte leaf records), nodes + leaf records staged into LDS per
                               workgroup (read through the vector caches when the image does not fit the 160 KB of LDS)   */
#define PBRT_ACCEL_BVH_GLOBAL 3u /* BVH4 read through the vector caches even if it would fit LDS (the kernel variant of
                               large meshes, forced: for tests and A/B runs)                 */

typedef struct pbrt_scene_desc {
    uint32_t n_prims;
    const pbrt_prim *prims;
    uint32_t n_materials;
    const pbrt_material *materials;
    uint32_t n_emitters;
    const pbrt_emitter *emitters;
    uint32_t n_light_prims;
    const uint32_t *light_prims;
    const float *light_cdf;
    uint32_t accel; /* PBRT_ACCEL_* */
    /* optional, [n_prims][9]: the vertex normals n0, n1, n2 (world space, unit length) of a primitive of a mesh that has
     * them: the shading normal of a hit is si.sh_frame.n = normalize(b0 n0 + b1 n1 + b2 n2) (Mitsuba Mesh), the geometric
     * normal g[9..11] stays the one rays are offset along.  A PARALLELOGRAM row holds one normal three times (a merged
     * quad keeps its vertex normals only if they all agree).  All-zero row, or NULL: the face normal. */
    const float *vertex_normals;
} pbrt_scene_desc;

/* ---- camera / film (radiance mode) -------------------------------------------------------- */
/* Mitsuba 'perspective' sensor (scenes/cbox.xml:11-21, scenes/simple.xml:7-12).
 * to_world: row-major 3x4 camera-to-world matrix; columns are (left, up, dir, origin).  */
typedef struct pbrt_camera {
    float to_world[12];
    float tan_half_fov_x;
    float near_clip;
    float far_clip;
    uint32_t film_w; /* full film size; crops are given per render call */
    uint32_t film_h;
} pbrt_camera;

#define PBRT_FILTER_BOX 0u      /* radius 0.5 (scenes/simple.xml:17) */
#define PBRT_FILTER_TENT 1u     /* radius 1.0 (scenes/cbox.xml:28)   */
#define PBRT_FILTER_GAUSSIAN 2u /* stddev 0.5, radius 2.0            */

typedef struct pbrt_film_desc {
    uint32_t crop_x, crop_y, crop_w, crop_h; /* output window inside the full film      */
    uint32_t spp;                            /* samples per pixel rendered by this call  */
    uint32_t sample_offset;                  /* index of the first sample (RNG key)      */
    uint32_t max_depth;                      /* Mitsuba path 'max_depth' (cbox.xml:4,8)  */
    uint32_t rr_depth;                       /* Mitsuba path 'rr_depth' (default 5)      */
    uint32_t filter;                         /* PBRT_FILTER_*                            */
    uint32_t seed;
    uint32_t flags;      /* PBRT_FILM_*                                                  */
    uint32_t pass_paths; /* 0 = library default; paths kept in flight per pass.  Default: 64 Mi for brute-force scenes
                            (8 B..120 B of workspace per path, by launch plan); BVH scenes (340 B per path) the largest power of
                            two, 1 .. 512 Mi, whose workspace fits two thirds of the free device memory -- pbrt_stats reports both */
} pbrt_film_desc;

#define PBRT_FILM_RAW_ACCUM 1u /* output 4 floats/pixel (sum w*rgb, sum w) un-normalised: \
                                  for sample-sharded multi-GPU reduction */
/* The flags marked DIAGNOSTIC BUILD select launch structures that lost their A/B: the product library does not carry their kernels
 * and refuses them with PBRT_E_UNSUPPORTED; libpbrt_hip_diag.so (make -C csrc diag, -DPBRT_DIAG) has them.  Same film either way. */
#define PBRT_FILM_NO_REPACK 2u /* DIAGNOSTIC BUILD, with PBRT_FILM_NO_HIT_POOL: do not re-densify the live paths before bounces >= 2 */
#define PBRT_FILM_FUSE_PLAN_SET 0x80u /* bits 8..15 of flags hold the fuse plan: bit d set = the launch that walks bounce d goes on \
                                        with bounce d + 1 of its paths in registers, up to 6 bounces per launch (brute-force \
                                        kernels; same film whatever the plan).  PBRT_FILM_FUSE_PLAN(0) = one launch per bounce, \
                                        (0x15) = pairs, (0xff) = six bounces per launch.  Unset: the library chooses from how many \
                                        paths survive each bounce (it goes on while >= 45 % do) -- measured on a 2-spp probe \
                                        pass at the start of the first render of a scene, afterwards on the scene's last render */
#define PBRT_FILM_FUSE_PLAN(mask) (PBRT_FILM_FUSE_PLAN_SET | (((mask) & 0xffu) << 8))
#define PBRT_FILM_WALK_SET 0x10000u /* DIAGNOSTIC BUILD: bits 17..24 of flags hold the depth from which ONE launch walks every remaining \
                                       bounce of a pass (k_walk, brute-force kernels: the workgroup that owns a segment carries its \
                                       survivors on; 0 = one launch per pass, 0xff = never) */
#define PBRT_FILM_WALK_FROM(d) (PBRT_FILM_WALK_SET | (((d) & 0xffu) << 17))
#define PBRT_FILM_REGEN 8u /* DIAGNOSTIC BUILD: brute-force scenes, persistent waves with path regeneration (k_regen: no path state in \
                             memory, one launch per pass; measured slower than the wavefront launches) */
#define PBRT_FILM_NO_HIT_POOL 0x10u /* DIAGNOSTIC BUILD: BVH scenes through the fused bounce kernel (closest hit, shading and shadow ray in \
                                    one launch per bounce, 4 waves per SIMD) instead of k_trace + k_shade (kernels_wavefront.h) */
#define PBRT_FILM_NO_OCCLUDER_PRUNING 4u /* diagnostic: next-event shadow segments of brute-force scenes walk EVERY primitive \
                                           instead of the occluder list (DESIGN D11: primitives on the scene's convex hull \
                                           and the lone area light are left out of it) -- same film if the pruning is right */

/* ---- ultrasound (acoustic) mode ----------------------------------------------------------- */
/* Parameter block of UltraIntegrator (CustomIntegrator.py:13-48) plus the sensor transform
 * the integrator reads from scene.sensors()[0].transform (CustomIntegrator.py:272). */
#define PBRT_US_MAX_ANGLES 64

/* CustomEmitter (CustomEmmitter.py:5-49), linear or convex array. */
typedef struct pbrt_us_emitter {
    uint32_t number_of_elements;
    float pitch, element_width, element_height;
    float radius;        /* 0: linear */
    float opening_angle; /* degrees   */
    uint32_t number_of_rays_per_element;
    float speed_of_sound;
    float steering_angle_min, steering_angle_max; /* degrees */
} pbrt_us_emitter;

/* Where the primary ray of a path comes from (pbrt_us_params.primary).
 * ELEMENT  the integrator's own deterministic ray, CustomIntegrator.py:264-273: origin T (x_e, 0, 0), direction
 *          normalize(T (sin theta_a, 0, cos theta_a)), emission time tx_delay[a, e], amplitude 1.  All paths of an (angle, element)
 *          pair share it (which is what the first-bounce tables exploit).
 * EMITTER  every path draws its own ray from CustomEmitter.sample_ray (CustomEmmitter.py:81-107, pbrt_us_params.emitter) -- what
 *          BASELINE config 3 "with CustomBSDF + CustomEmmitter" means read literally.  The reference never connects the two classes
 *          (its integrator does not call the emitter), so the connection is a definition of this library: the
 *          (angle, element) grid of the acquisition stratifies the emitter's sample space -- path k of pair (a, e) calls
 *              sample_ray(time = 0, sample1 = (e + 1/2) / N, sample2 = (u.x, u.y), sample3 = (a + u.z) / n_angles)
 *          with u = the path's RNG block 0x80000000, i.e. element e (jittered inside the element by sample2, :64-68) and a steering
 *          angle drawn uniformly from the a-th of n_angles equal parts of [steering_angle_min, steering_angle_max] (:85-87); the
 *          ray is taken to the world with the sensor transform like the integrator's own (:272-273), its `time` (the element's
 *          steering delay -x sin(psi) / c, :93-94) is the path's initial time of flight (t0 of :329 is then 0) and its weight
 *          max(0, d.n) / N_total_rays (:97-98) multiplies every echo the path deposits -- as Mitsuba's render loop multiplies what
 *          Integrator.sample returns by the weight of the sampled ray; the path's own amplitude starts at 1 (:276), so the
 *          Russian roulette of :364-367, which reads that amplitude as a survival probability, is the integrator's.
 *          emitter.number_of_elements must equal n_elements.  No first-bounce tables: every path walks the scene from its own
 *          origin. */
#define PBRT_US_PRIMARY_ELEMENT 0u
#define PBRT_US_PRIMARY_EMITTER 1u

typedef struct pbrt_us_params {
    uint32_t max_depth;    /* CustomIntegrator.py:16 */
    float frequency;       /* :17 */
    float sound_speed;     /* :18 */
    float attenuation;     /* :19 */
    float main_beam_angle; /* degrees, :21 */
    float cutoff_angle;    /* degrees, :22 */
    float fs;              /* 'sampling_rate', :23 */
    uint32_t n_elements;   /* :26 */
    float pitch;           /* :27 */
    uint32_t n_angles;     /* :33-34 */
    float angles_deg[PBRT_US_MAX_ANGLES];
    uint32_t time_samples; /* :42 */
    float sensor_to_world[12]; /* row-major 3x4 */
    float max_path_len;        /* hard-coded 0.2 in the reference (:307,372) */
    uint32_t quirks;           /* PBRT_USQ_* */
    uint32_t primary;          /* PBRT_US_PRIMARY_* (ABI 5) */
    pbrt_us_emitter emitter;   /* read when primary == PBRT_US_PRIMARY_EMITTER */
} pbrt_us_params;

/* Behaviour switches; each bit reproduces one reference quirk.
 * The library default (quirks = PBRT_USQ_REFERENCE) is the literal arithmetic of the scalar
 * variant simulate_acquisition_parallel (CustomIntegrator.py:235-376) and UltraBSDF.sample
 * (CustomBSDF.py:87-175) with ONE repair: 'survive' is initialised to True (quirk B5), because
 * as checked in the loop raises UnboundLocalError on the first surviving bounce. */
#define PBRT_USQ_DIAG_SAMPLE 0x1u   /* A2: scalar sample broadcast to Point2f(s,s)             */
#define PBRT_USQ_REF_REFLECT 0x2u   /* A5: reflected = wi + 2 cos m (not the mirror direction) */
#define PBRT_USQ_UNIT_GGX_PDF 0x4u  /* A7: ggx_pdf == 1.0                                       */
#define PBRT_USQ_DOUBLE_LOCAL 0x8u  /* A1: Frame(n).to_local applied to the already-local wi   */
#define PBRT_USQ_MIXED_FRAMES 0x10u /* A8: world shading normal dotted with local wi           */
#define PBRT_USQ_NEVER_ENTER 0x20u  /* A4: 'entering' is always false                          */
#define PBRT_USQ_CLAMP_TIME 0x40u   /* B3 (Dr.Jit variant): clamp t_idx instead of dropping    */
#define PBRT_USQ_NO_TOF_ACCUM 0x80u /* B2 (Dr.Jit variant): tof never accumulates              */
#define PBRT_USQ_NO_CARRIER 0x100u  /* f-3 pulse model: deposit atten*amp*w_i*w_o without the sin(phase) factor of :348; the carrier
                                       comes from pbrt_us_apply_pulse (not a reference quirk)  */
#define PBRT_USQ_NO_FIRST_TABLES 0x200u /* diagnostic: every path walks the scene at its first bounce (same result) */
#define PBRT_USQ_NO_FUSED_BOUNCES 0x400u /* diagnostic: one launch per bounce instead of one per pass */
#define PBRT_USQ_FROZEN_DRAWS 0x800u  /* B1 (Dr.Jit variant): np.random.uniform is evaluated while dr.while_loop TRACES its body
                                         (CustomIntegrator.py:153,173-174,219), so every bounce of a ray reuses the draws of its
                                         first one: the RNG block of bounce b is block 0                              */
#define PBRT_USQ_SIGNED_RR 0x1000u    /* B5 (Dr.Jit variant, :219-224): rr_prob = min(atten * amp, 1) without the abs, survive
                                         iff u < rr_prob (strict), atten = survive ? atten / rr_prob : 0             */
/* the Dr.Jit variant simulate_acquisition (CustomIntegrator.py:60-232), on top of PBRT_USQ_REFERENCE */
#define PBRT_USQ_DRJIT_VARIANT (PBRT_USQ_CLAMP_TIME | PBRT_USQ_NO_TOF_ACCUM | PBRT_USQ_FROZEN_DRAWS | PBRT_USQ_SIGNED_RR)
#define PBRT_USQ_REFERENCE                                                                \
    (PBRT_USQ_DIAG_SAMPLE | PBRT_USQ_REF_REFLECT | PBRT_USQ_UNIT_GGX_PDF | PBRT_USQ_DOUBLE_LOCAL | \
     PBRT_USQ_MIXED_FRAMES | PBRT_USQ_NEVER_ENTER)

/* UltraSensor (bytecode-only class; props used at USMain.py:43-65 and
 * MitsubaScenes/Sphere_Box.xml:16-34). */
typedef struct pbrt_us_sensor {
    uint32_t num_elements;
    float element_width, element_height, pitch;
    float radius; /* +inf: linear array */
    float center_frequency, sound_speed, directivity;
    float to_world[12];
} pbrt_us_sensor;

/* CustomSensor.put_data accumulator (CustomSensor.py:7-59). */
typedef struct pbrt_us_receiver {
    uint32_t number_of_elements;
    float pitch;
    float sample_rate;
    uint32_t time_samples;
} pbrt_us_receiver;

typedef struct pbrt_stats {
    uint64_t samples;      /* camera / transducer paths started by the last call           */
    uint64_t segments;     /* path segments shaded (sum over bounces of live paths)         */
    uint64_t shadow_rays;  /* occlusion rays                                               */
    double kernel_ms;      /* device time of the last call, HIP events on the ctx stream    */
    double bounce_ms;      /* of which: the dominant (bounce) kernels                       */
    uint32_t bounce_launches;
    uint32_t passes;
    uint64_t model_bytes;  /* algorithmic HBM bytes of the last call (DESIGN.md byte model) */
    uint64_t bounce_model_bytes;
    uint64_t live[16];     /* paths entering depth d (d = 0..15), summed over passes          */
    /* how the last call was launched (ABI 3).  The launch plan of a brute-force scene is state of the pbrt_scene: unless the
     * caller sets one (PBRT_FILM_FUSE_PLAN) it is learnt from the path survival of the scene's previous render, so TIMINGS depend
     * on the call history of a scene -- films never do. */
    uint32_t fuse_plan;    /* bit d: the launch that walks bounce d goes on with bounce d + 1 (brute-force scenes; 0 for BVH scenes) */
    uint32_t plan_source;  /* PBRT_PLAN_* */
    uint64_t pass_paths;   /* paths in flight per pass */
    uint64_t workspace_bytes; /* device memory the context holds after the call */
    uint64_t trace_model_bytes; /* ABI 4, BVH scenes: the part of bounce_model_bytes that belongs to the k_trace launches (the rest is k_shade's) */
} pbrt_stats;
#define PBRT_PLAN_CALLER 0u  /* pbrt_film_desc.flags carried PBRT_FILM_FUSE_PLAN */
#define PBRT_PLAN_LEARNT 1u  /* from the path survival of the scene's previous render */
#define PBRT_PLAN_PROBED 2u  /* first render of the scene: from a 2-spp probe pass at the start of this call */
#define PBRT_PLAN_DEFAULT 3u /* first render, too few samples for a probe: the library default (pairs) */
#define PBRT_PLAN_STREAMS 4u /* BVH scenes: one k_trace + one k_shade launch per bounce */

typedef struct pbrt_ctx pbrt_ctx;
typedef struct pbrt_scene pbrt_scene;

/* ---- context / scene ---------------------------------------------------------------------- */
int pbrt_abi_version(void);
/* replaces: mi.set_variant(...) (USMain.py:12) -- selects the device instead of a JIT backend */
int pbrt_ctx_create(int device, pbrt_ctx **out);
int pbrt_ctx_destroy(pbrt_ctx *ctx);
const char *pbrt_last_error(pbrt_ctx *ctx); /* ctx may be NULL: last ctx-less error */
int pbrt_get_stats(pbrt_ctx *ctx, pbrt_stats *out);
/* Workspace policy (ABI 4).  A context keeps the device buffers of its calls (path state, ray and radiance records, film
 * accumulators) and re-uses them; pbrt_stats.workspace_bytes says how much it holds.  BVH scenes size a pass to the largest power
 * of two of paths (1 .. 512 Mi, 340 B each) that fits two thirds of the free device memory -- up to 183 GB on an idle MI355X
 * (larger passes are faster: their twelve launches per pass fill the chip longer).  A caller
 * that shares the device (the reference's driver imports torch beside the renderer, USMain.py:5) bounds that with a limit -- here,
 * or PBRT_WORKSPACE_LIMIT_BYTES in the environment at pbrt_ctx_create; 0 = none -- and hands memory back with pbrt_ctx_trim.
 * Under a limit, and when an allocation fails, renders take smaller passes (same film, more passes); a request that cannot be met
 * at the smallest pass returns PBRT_E_NOMEM.  pbrt_film_desc.pass_paths still overrides the choice per call. */
int pbrt_ctx_set_workspace_limit(pbrt_ctx *ctx, uint64_t bytes); /* a limit below what the context holds frees its buffers at once */
/* frees every workspace buffer the most recent call did not use and every one larger than that call needed; *held_after
 * (may be NULL) = bytes still held */
int pbrt_ctx_trim(pbrt_ctx *ctx, uint64_t *held_after);

/* replaces: mi.load_dict / mi.load_file scene instantiation + accel build (USMain.py:257) */
int pbrt_scene_create(pbrt_ctx *ctx, const pbrt_scene_desc *desc, pbrt_scene **out);
/* replaces: params[key] = v; params.update() (USMain.py:264-265) -> BSDF.parameters_changed
 * (CustomBSDF.py:186-191).  Overwrites material `index` in place, no accel rebuild. */
int pbrt_scene_update_material(pbrt_scene *scene, uint32_t index, const pbrt_material *m);
int pbrt_scene_destroy(pbrt_scene *scene);

/* ---- the hot path: radiance mode ----------------------------------------------------------- */
/* replaces: mi.render(scene) == SamplingIntegrator::render -> Sensor.sample_ray ->
 * Integrator.sample (CustomIntegrator.py:52-53 is the stub; semantics SURVEY.md App. D) ->
 * film.  out_rgb: host float[crop_h*crop_w*3] (or *4 with PBRT_FILM_RAW_ACCUM). */
int pbrt_render_radiance(pbrt_scene *scene, const pbrt_camera *cam, const pbrt_film_desc *film, float *out_rgb);
/* same, output left in HBM at device pointer d_out (no PCIe copy) */
int pbrt_render_radiance_dev(pbrt_scene *scene, const pbrt_camera *cam, const pbrt_film_desc *film, void *d_out);

/* replaces: Integrator.sample(scene, sampler, ray, medium, active) -> (spec, mask, aovs)
 * (CustomIntegrator.py:52-53 is a stub returning 0; radiance semantics SURVEY.md App. D).
 * Radiance arriving along n caller-supplied rays: o,d [3][n] (unit d), tmax [n]; the RNG key of ray i
 * is (index_offset + i, sample_index).  out rgb [3][n]. */
int pbrt_integrator_sample(pbrt_scene *scene, uint32_t n, const float *o, const float *d, const float *tmax,
                           uint32_t index_offset, uint32_t sample_index, uint32_t seed, uint32_t max_depth,
                           uint32_t rr_depth, float *rgb);

/* ---- the hot path: ultrasound mode ---------------------------------------------------------- */
/* replaces: UltraIntegrator.simulate_acquisition_parallel(scene) (CustomIntegrator.py:235-405),
 * called from USMain.py:99.  Traces paths k in [path_offset, path_offset+paths_per_ray) for
 * every (angle, element) pair; channel_buf[(a*n_elements+recv)*time_samples + t] accumulates the
 * echoes divided by norm_paths (pass the TOTAL paths per ray of the job so that shards sum to
 * the single-device result); tx_delays[a*n_elements+e] as CustomIntegrator.py:254-257. */
int pbrt_us_acquire(pbrt_scene *scene, const pbrt_us_params *p, uint32_t seed, uint32_t paths_per_ray,
                    uint32_t path_offset, uint32_t norm_paths, float *channel_buf, float *tx_delays);
int pbrt_us_acquire_dev(pbrt_scene *scene, const pbrt_us_params *p, uint32_t seed, uint32_t paths_per_ray,
                        uint32_t path_offset, uint32_t norm_paths, void *d_channel_buf, float *tx_delays);

/* ---- batched leaf operators (host SoA in/out) ---------------------------------------------- */
/* replaces: scene.ray_intersect(ray) (CustomIntegrator.py:146,309).  o,d: [3][n]; tmax: [n];
 * out t [n] (+inf: miss), prim [n] (0xffffffff: miss), u,v [n]. */
int pbrt_ray_intersect(pbrt_scene *scene, uint32_t n, const float *o, const float *d, const float *tmax, float *t,
                       uint32_t *prim, float *u, float *v);
/* replaces: scene.ray_intersect(si.spawn_ray(sec_dir)).is_valid() used as an occlusion test
 * (CustomIntegrator.py:159-160,324-325).  hit [n] is 0/1. */
int pbrt_ray_test(pbrt_scene *scene, uint32_t n, const float *o, const float *d, const float *tmax, uint8_t *hit);

/* replaces: BSDF.sample(ctx, si, sample1, sample2) (CustomBSDF.py:87-175 for ULTRA; Mitsuba
 * diffuse / conductor / dielectric per SURVEY.md App. D).
 *  wi      [3][n] incident direction in the local shading frame (si.wi)
 *  n_geo   [3][n] world geometric normal (si.n)        -- read by ULTRA only
 *  n_sh    [3][n] world shading normal (si.sh_frame.n) -- read by ULTRA only
 *  sh_s    [3][n] tangent of the shading frame (si.sh_frame.s, or the shape's dp_du: Mitsuba builds the frame as
 *                 s = normalize(dp_du - n (n . dp_du)), t = n x s) -- read by ULTRA only, for bs.wo = si.to_local(chosen)
 *                 (CustomBSDF.py:165); NULL: coordinate_system(n_sh)
 *  s1 [n], s2 [2][n] the uniform variates
 *  out: wo [3][n] local, pdf [n], weight [3][n] (ULTRA: amplitude in weight[0], rest equal),
 *       sampled [n]: 0 = reflection lobe, 1 = transmission lobe, 0xffffffff = invalid sample */
int pbrt_bsdf_sample(pbrt_ctx *ctx, const pbrt_material *m, uint32_t quirks, uint32_t n, const float *wi,
                     const float *n_geo, const float *n_sh, const float *sh_s, const float *s1, const float *s2, float *wo,
                     float *pdf, float *weight, uint32_t *sampled);
/* replaces: BSDF.eval / BSDF.pdf / BSDF.eval_pdf (CustomBSDF.py:177-184: constant 0 for ULTRA).
 * f [3][n] = bsdf value * cos(theta_o). */
int pbrt_bsdf_eval_pdf(pbrt_ctx *ctx, const pbrt_material *m, uint32_t n, const float *wi, const float *wo, float *f,
                       float *pdf);

/* Emitter.sample_direction(it, sample) -- required by the north star, absent from the
 * reference; semantics of Mitsuba area / point emitters.  No visibility test.
 *  p [3][n] reference points; u [4][n] variates (emitter pick, primitive pick, 2-D position)
 *  out: d [3][n], dist [n], pdf [n] (solid angle; 1 for delta), weight [3][n] = radiance/pdf,
 *       q [3][n] sampled point, emitter [n] */
int pbrt_emitter_sample_direction(pbrt_scene *scene, uint32_t n, const float *p, const float *u, float *d, float *dist,
                                  float *pdf, float *weight, float *q, uint32_t *emitter);

/* replaces: Sensor.sample_ray of the Mitsuba 'perspective' sensor. pos [2][n] in [0,1)^2 over
 * the full film.  out: o,d [3][n], tmax [n]. */
int pbrt_sensor_sample_ray(pbrt_ctx *ctx, const pbrt_camera *cam, uint32_t n, const float *pos, float *o, float *d,
                           float *tmax);

/* replaces: UltraSensor.sample_ray(time, wavelength_sample, position_sample, aperture_sample)
 * (SURVEY.md App. C).  time, wavelength_sample [n]; position_sample, aperture_sample [2][n].
 * use_hemisphere_warp != 0 selects the warp.square_to_uniform_hemisphere branch. */
int pbrt_us_sensor_sample_ray(pbrt_ctx *ctx, const pbrt_us_sensor *s, int use_hemisphere_warp, uint32_t n,
                              const float *time, const float *wavelength_sample, const float *position_sample,
                              const float *aperture_sample, float *o, float *d, float *weight);

/* replaces: CustomEmitter.sample_ray(time, sample1, sample2, sample3) (CustomEmmitter.py:81-107)
 * incl. sample_position (:51-79).  time, s1, s3 [n]; s2 [2][n].
 * out: o,d [3][n], ray_time [n], weight [n], pdf_pos [n]. */
int pbrt_us_emitter_sample_ray(pbrt_ctx *ctx, const pbrt_us_emitter *e, uint32_t n, const float *time,
                               const float *s1, const float *s2, const float *s3, float *o, float *d, float *ray_time,
                               float *weight, float *pdf_pos);

/* replaces: CustomSensor.put_data(ray, amplitude) (CustomSensor.py:29-59), batched: ray origin
 * x [n], ray time [n], ray direction [3][n], amplitude [n] accumulated into
 * channel_buffer[number_of_elements*time_samples] (host, read-modify-write). */
int pbrt_us_put_data(pbrt_ctx *ctx, const pbrt_us_receiver *r, uint32_t n, const float *ox, const float *time,
                     const float *d, const float *amplitude, float *channel_buffer);

/* tx_delay[a,e] = elem_x[e] * sin(theta_a) / c  (CustomIntegrator.py:246-257); host-only helper
 * shared by pbrt_us_acquire. */
int pbrt_us_tx_delays(const pbrt_us_params *p, float *tx_delays);

/* ---- image formation behind the hot path ------------------------------------------
 * The reference hands the channel buffer to the third-party `ultraspy` package (absent here; parity unpinned):
 * DelayAndSum.beamform(data, GridScan(x, z)) -> compute_envelope -> manual log compression
 * (USMain.py:126-221).  These entry points are this build's own definition of those three steps. */
#define PBRT_DAS_NEAREST 0u
#define PBRT_DAS_LINEAR 1u
typedef struct pbrt_das_params {
    uint32_t n_angles, n_elements, time_samples; /* data [n_angles][n_elements][time_samples] (RF, f32) */
    float fs, sound_speed, t0;                   /* sample s was taken at t0 + s / fs */
    float f_number;                              /* receive element e used iff |x - x_e| <= z / (2 f_number); 0: all */
    uint32_t interpolation;                      /* PBRT_DAS_NEAREST / PBRT_DAS_LINEAR */
    uint32_t compound_mean;                      /* 0: sum over transmissions, 1: mean */
    uint32_t nx, nz;                             /* GridScan(x, z) */
} pbrt_das_params;

/* replaces: ultraspy DelayAndSum.beamform(d_data, scan) as called at USMain.py:204.
 * out[ix][iz] = sum_a sum_e data[a][e](t_tx(a; x, z) + |(x, z) - (elem_x[e], 0)| / c), where the transmit time is the
 * first arrival of the emitted wavefront, t_tx = min_e' (tx_delays[a][e'] + |(x, z) - (elem_x[e'], 0)| / c)  (equal to
 * (x sin(theta) + z cos(theta)) / c for the plane-wave delays of pbrt_us_tx_delays inside the aperture's shadow).
 * Sample positions are evaluated in f64, samples interpolated and summed in f32 in the order (groups of 8 transmissions, element,
 * transmission within the group).  Host pointers. */
int pbrt_das_beamform(pbrt_ctx *ctx, const pbrt_das_params *p, const float *data, const float *tx_delays,
                      const float *elem_x, const float *x, const float *z, float *out);

/* replaces: ultraspy DelayAndSum.compute_envelope(d_output, scan) on RF data (USMain.py:205): modulus of the
 * analytic signal along the axial (z, fastest) axis, i.e. |scipy.signal.hilbert(rf, axis=-1)|.  nz <= 4096. */
int pbrt_envelope(pbrt_ctx *ctx, uint32_t nx, uint32_t nz, const float *rf, float *env);

/* replaces: the manual log compression of USMain.py:210-218: db = 20 log10(env + 1e-12), clipped to
 * [max(db) - dynamic_range_db, max(db)], mapped to [0, 1].  n values in, n values out. */
int pbrt_log_compress(pbrt_ctx *ctx, uint32_t n, const float *env, float dynamic_range_db, float *out);

/* SURVEY.md section 8 f-3, the pulse model of the reference's prototype (RayTracingV0.py:194-204, "UltraRay Eq. 14"):
 *   pulse(t; t0, amp) = amp * sin(2 pi fc (t - t0)) * exp(-(t - t0)^2 / sigma^2).
 * With echoes deposited as plain amplitudes on the sample grid (PBRT_USQ_NO_CARRIER) the RF trace is the discrete
 * convolution of every trace with h[k] = sin(2 pi fc k / fs) * exp(-(k / fs)^2 / sigma^2), |k| <= ceil(2.5 sigma fs):
 * out[tr][n] = sum_k in[tr][n - k] * h[k]  (zero outside the trace).  n_traces x time_samples values, host pointers;
 * in and out must not overlap.  [DEFINE] sigma = wave_cycles / (4 fc) turns the integrator's unused `wave_cycles`
 * (CustomIntegrator.py:20) into the pulse length. */
int pbrt_us_apply_pulse(pbrt_ctx *ctx, uint32_t n_traces, uint32_t time_samples, float fs, float frequency, float sigma,
                        const float *in, float *out);

/* ---- ABI 5: the reference's us_render loop without leaving HBM -------------------------------------------------------
 * USMain.py:92-252 runs acquisition -> DAS -> envelope -> log compression 51 times per script (:260, :279-283).  With the
 * host-pointer forms above every step crosses PCIe twice (the 12.8 MB channel buffer up, an image down).  The *_dev forms take
 * device pointers, queue their kernels on the context's stream and return without synchronising:
 *     pbrt_us_acquire_queue_dev(scene, ..., d_channel, tx)      queued (pbrt_us_acquire_dev: the same, and waits)
 *     [pbrt_us_apply_pulse_dev(ctx, ..., d_channel, d_rf)]       pulse_model = "gaussian" only
 *     pbrt_das_beamform_dev(ctx, &das, d_rf, d_tx, d_elem_x, d_x, d_z, d_bf)
 *     pbrt_envelope_dev(ctx, nx, nz, d_bf, d_env)
 *     pbrt_log_compress_dev(ctx, nx * nz, d_env, 60, d_img)
 *     pbrt_dev_download(ctx, img, d_img, nx * nz * 4)            the ONE copy to the host; waits for the stream
 * Same kernels, same results bit for bit as the host-pointer forms (which stage their arguments and call the same code). */
/* pbrt_us_acquire_dev without the wait: the acquisition is queued on the context's stream and the call returns; the channel
 * buffer is complete for whatever is queued behind it (the image-formation *_dev calls) and for the host after
 * pbrt_ctx_synchronize / pbrt_dev_download.  Its statistics -- and the error of a tripped traversal guard, PBRT_E_DEVICE --
 * arrive with the next call on the context that waits for the stream or starts other work (pbrt_get_stats, pbrt_ctx_synchronize,
 * pbrt_dev_download, the next acquisition or render ...): that call finishes the queued acquisition first and returns its error.
 * tx_delays (host, may be NULL) is filled before the call returns. */
int pbrt_us_acquire_queue_dev(pbrt_scene *scene, const pbrt_us_params *p, uint32_t seed, uint32_t paths_per_ray,
                              uint32_t path_offset, uint32_t norm_paths, void *d_channel_buf, float *tx_delays);

/* replaces: ultraspy DelayAndSum.beamform(d_data, scan) (USMain.py:204), data and tables in HBM.  All pointers are device
 * pointers: d_data [n_angles][n_elements][time_samples], d_tx_delays [n_angles][n_elements], d_elem_x [n_elements],
 * d_x [nx], d_z [nz], d_out [nx][nz]. */
int pbrt_das_beamform_dev(pbrt_ctx *ctx, const pbrt_das_params *p, const void *d_data, const void *d_tx_delays,
                          const void *d_elem_x, const void *d_x, const void *d_z, void *d_out);
/* The transmit half of the delays depends on the transmit delays and the scan grid only, and the reference's loop changes neither
 * (one probe, one GridScan, 51 renders: USMain.py:129-204): pbrt_das_first_arrival_dev writes the table
 *     d_table[a][ix][iz] (double) = t_tx(a; x, z) = min_e' (tx_delays[a][e'] + |(x, z) - (elem_x[e'], 0)| / c)
 * once, pbrt_das_beamform_table_dev beamforms with it: the same image bit for bit as pbrt_das_beamform_dev (which evaluates that
 * minimum per call), without the pass over all elements.  Both queue their kernel on the context's stream and return. */
int pbrt_das_first_arrival_dev(pbrt_ctx *ctx, const pbrt_das_params *p, const void *d_tx_delays, const void *d_elem_x,
                               const void *d_x, const void *d_z, void *d_table);
int pbrt_das_beamform_table_dev(pbrt_ctx *ctx, const pbrt_das_params *p, const void *d_data, const void *d_table,
                                const void *d_elem_x, const void *d_x, const void *d_z, void *d_out);
/* replaces: DelayAndSum.compute_envelope (USMain.py:205); d_rf, d_env [nx][nz], distinct buffers */
int pbrt_envelope_dev(pbrt_ctx *ctx, uint32_t nx, uint32_t nz, const void *d_rf, void *d_env);
/* replaces: the log compression of USMain.py:210-218; d_env, d_out [n] (may be the same buffer) */
int pbrt_log_compress_dev(pbrt_ctx *ctx, uint32_t n, const void *d_env, float dynamic_range_db, void *d_out);
/* the pulse model (RayTracingV0.py:194-204) on a channel buffer in HBM; d_in, d_out [n_traces][time_samples], distinct */
int pbrt_us_apply_pulse_dev(pbrt_ctx *ctx, uint32_t n_traces, uint32_t time_samples, float fs, float frequency, float sigma,
                            const void *d_in, void *d_out);

/* waits for everything queued on the context's stream */
int pbrt_ctx_synchronize(pbrt_ctx *ctx);
/* Device buffers for a caller without a GPU library of its own (the reference's driver is NumPy: USMain.py:103-121).
 * pbrt_dev_upload copies in stream order (a pageable source is staged before it returns: the caller may reuse it at once);
 * pbrt_dev_download copies in stream order and waits; pbrt_dev_free waits for the stream first. */
int pbrt_dev_alloc(pbrt_ctx *ctx, uint64_t bytes, void **out);
int pbrt_dev_free(pbrt_ctx *ctx, void *p);
int pbrt_dev_upload(pbrt_ctx *ctx, void *dst_dev, const void *src_host, uint64_t bytes);
int pbrt_dev_download(pbrt_ctx *ctx, void *dst_host, const void *src_dev, uint64_t bytes);

/* Per-step device times of the image-formation entry points (host and *_dev forms alike), HIP events on the ctx stream.
 * Off by default (an event pair per step costs a few microseconds of queue time each); pbrt_ctx_set_profiling(ctx, 1) turns
 * them on, pbrt_get_image_stats waits for the stream and reports the most recent call of each step since then. */
typedef struct pbrt_image_stats {
    double pulse_ms, das_ms, envelope_ms, log_ms;
    uint64_t das_model_bytes; /* algorithmic bytes of the last DAS call: the channel buffer once + the image once */
    uint32_t measured;        /* bit 0 pulse, 1 das, 2 envelope, 3 log: steps that ran with profiling on */
    uint32_t pad;
} pbrt_image_stats;
int pbrt_ctx_set_profiling(pbrt_ctx *ctx, int on);
int pbrt_get_image_stats(pbrt_ctx *ctx, pbrt_image_stats *out);

/* ---- the queued chain as ONE submission (hipGraph) ---------------------------------------------------------------------
 * At the reference's own size (320 rays x 1 path, USMain.py:36) the chain above is eight small kernels, three fills and a copy:
 * the host spends as long queueing them as the device spends running them, 100 times per script.  Between
 * pbrt_ctx_record_begin and pbrt_ctx_record_end the queueing entry points (pbrt_us_acquire_queue_dev, pbrt_us_apply_pulse_dev,
 * pbrt_das_beamform_dev, pbrt_das_beamform_table_dev, pbrt_envelope_dev, pbrt_log_compress_dev) are RECORDED on the context's
 * stream instead of run; pbrt_graph_launch replays the recording in one submission and returns without waiting, exactly as if
 * the recorded calls had just been made: same kernels, same arguments, same results bit for bit, the acquisition's statistics
 * arrive with the next call that waits (kernel_ms / bounce_ms are 0: a replay carries no event pairs).
 *  - Arguments are frozen at recording time: parameter blocks, seeds, path counts and every device pointer.  What the kernels
 *    READ through those pointers is not: pbrt_scene_update_material between two launches is seen by the second one (the
 *    finite-difference loop of USMain.py:262-289 changes one roughness per render and nothing else).
 *  - The workspace must be warm: run the chain once the ordinary way first; a recorded call that would have to allocate, or to
 *    upload a table, fails (PBRT_E_NOMEM / PBRT_E_INVALID; the message names the buffer or table).  An error of a recorded
 *    call leaves the recording open: close it with pbrt_ctx_record_end, which reports the error again and makes no graph.
 *    Any other entry point on the context is refused with PBRT_E_INVALID while a recording is open (the recording stays good).
 *  - A recording goes stale when the context frees or replaces memory it refers to (a larger request for the same workspace
 *    buffer, pbrt_ctx_trim, pbrt_dev_free, pbrt_scene_destroy) or uploads other acquisition tables: pbrt_graph_launch then
 *    returns PBRT_E_INVALID and launches nothing; record again. */
typedef struct pbrt_graph pbrt_graph;
int pbrt_ctx_record_begin(pbrt_ctx *ctx);
int pbrt_ctx_record_end(pbrt_ctx *ctx, pbrt_graph **out);
int pbrt_graph_launch(pbrt_graph *g);
/* waits for the context's stream (a replay may still run).  Destroy a context's recordings before the context. */
int pbrt_graph_destroy(pbrt_graph *g);

#ifdef __cplusplus
}
#endif
#endif /* PBRT_HIP_H */
