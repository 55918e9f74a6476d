// pbrt_api.hip -- host side of libpbrt_hip.so: the C-ABI of include/pbrt_hip.h.
// One pbrt_ctx per device (HIP stream, grow-only workspace in HBM, last-error string); scenes are
// uploaded once and stay resident; every entry point is synchronous on return.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "bvh_build.h"
#include "kernels_us.h"
#include "kernels_wavefront.h"
#include "kernels_us_wavefront.h"
#include "kernels_beamform.h"

static std::string g_ctxless_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;   // allocated
    size_t need = 0;    // what the most recent request asked for
    uint64_t stamp = 0; // pbrt_ctx::call_seq of that request
};

// the context's pinned host page: partial sums of the ultrasound counters (written by k_us_reduce_stats itself), then the guard words
#define PIN_GUARD_OFFSET ((2 + MAX_DEPTH_STATS) * REDUCE_SLICES * 8)
#define PIN_BYTES (PIN_GUARD_OFFSET + 4096)

struct pbrt_ctx {
    int device = 0;
    int n_cu = 256;  // compute units (MI355X: 256); read from the device properties
    hipStream_t stream = nullptr;
    hipStream_t st_trace = nullptr, st_shade = nullptr;  // BVH scenes, PBRT_WF_SPLIT: streams with CU masks (wf_split_streams)
    uint32_t split_s = 0;
    std::vector<hipEvent_t> sync_ev;
    std::string err;
    pbrt_stats stats{};
    // Workspace: named device buffers that grow on demand and are re-used by later calls.  ws_limit (0: none) caps their sum --
    // PBRT_WORKSPACE_LIMIT_BYTES at pbrt_ctx_create, pbrt_ctx_set_workspace_limit later; a request that would exceed it fails with
    // PBRT_E_NOMEM and the render paths answer by taking smaller passes -- and pbrt_ctx_trim gives back what the last call did
    // not need (a caller that shares the device with another allocator, e.g. torch beside the renderer as in USMain.py:5).
    std::map<std::string, DevBuf> ws;
    size_t ws_limit = 0;
    uint64_t call_seq = 0;  // bumped by every entry point that takes workspace
    size_t ws_total() const {
        size_t t = 0;
        for (const auto &kv : ws) t += kv.second.bytes;
        return t;
    }
    void release(const char *name) {
        auto it = ws.find(name);
        if (it == ws.end()) return;
        if (it->second.p) {
            (void)hipFree(it->second.p);
            ++ws_epoch;
        }
        ws.erase(it);
    }
    std::vector<hipEvent_t> ev_pool;
    // An acquisition that was queued without waiting (pbrt_us_acquire_queue_dev, ABI 5): what us_finish needs to turn the counters
    // the device leaves in the pinned page into pbrt_stats once the stream has drained.  Every entry point that waits for the
    // stream or starts other work on the context finishes it first (ctx_settle).
    struct PendingAcq {
        bool active = false, streams = false, tab0 = false;
        bool scaled = true;  // the normalisation pass over the channel buffer ran (not at one path per ray)
        bool timed = true;  // false: replayed from a recording (no event pairs; kernel_ms / bounce_ms are 0)
        size_t n_ev = 0;
        uint32_t passes = 0, launches = 0;
        uint64_t samples = 0, nchan = 0;
    } pend;
    // one page of pinned host memory: the statistics and guard words of a call are copied here (a copy to pageable memory blocks
    // the host until it is done; to pinned memory it is queued like a kernel)
    void *pinned = nullptr;
    unsigned long long *pin_stats() { return (unsigned long long *)pinned; }              // [2 + MAX_DEPTH_STATS][REDUCE_SLICES] partial sums
    uint32_t *pin_guard() { return (uint32_t *)((char *)pinned + PIN_GUARD_OFFSET); }      // [WF_GUARD_WORDS]
    // the ultrasound counters (workspace "us_stats") as the last acquisition left them: k_us_reduce_stats zeroes what it reads, so a
    // call that finds the same buffer and this flag set skips the fill command
    const void *us_rows_clean = nullptr;
    size_t us_rows_clean_bytes = 0;
    uint64_t us_rows_epoch = 0;  // ws_epoch when the rows were left clean: a buffer freed since may have come back at the same address
    // the small tables of the last acquisition (transmit delays, primary directions, element positions) as uploaded: the
    // reference's loop calls the acquisition 51 times with the same ones (USMain.py:260,279-283), three host-to-device copies each
    std::vector<float> us_tab_host;
    const void *us_tab_dev = nullptr;
    // image formation (f-1): event pairs per step when profiling is on (pbrt_ctx_set_profiling), read by pbrt_get_image_stats
    bool profiling = false;
    hipEvent_t img_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t img_mask = 0;
    uint64_t img_das_bytes = 0;
    size_t env_lds_attr = 0;
    uint32_t env_taps_n = 0;  // column length the context's tap table (workspace "env_taps") was made for
    bool img_event(int i) { return img_ev[i] || hipEventCreate(&img_ev[i]) == hipSuccess; }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint32_t lds_limit = 0;
    // A recording of the queued chain (pbrt_ctx_record_begin .. pbrt_ctx_record_end; replayed by pbrt_graph_launch): while it is
    // open the context's stream captures instead of running, nothing may wait for it, allocate or upload.  ws_epoch counts the
    // events that make a finished recording stale: memory it may refer to freed or replaced, other acquisition tables uploaded,
    // the envelope's tap table made for another column length.
    bool recording = false, rec_failed = false;
    std::string rec_msg;
    uint64_t ws_epoch = 0;
    uint32_t n_graphs = 0;

    int fail(int code, const char *fmt, ...) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        if (recording && !rec_failed) {  // the first error inside a recording: pbrt_ctx_record_end reports it again and makes no graph
            rec_failed = true;
            rec_msg = buf;
        }
        return code;
    }
    // an entry point that may not run while the stream records: the same, but the recording stays good
    int refuse(const char *what) {
        err = std::string("a recording is open on this context (pbrt_ctx_record_begin): ") + what;
        return PBRT_E_INVALID;
    }
    // returns nullptr on failure (err set)
    void *buf(const char *name, size_t bytes) {
        DevBuf &b = ws[name];
        b.need = bytes;
        b.stamp = call_seq;
        if (b.bytes >= bytes && b.p) return b.p;
        if (recording) {  // (the caller reports PBRT_E_NOMEM with this message)
            fail(PBRT_E_NOMEM, "recording: workspace buffer %s (%zu bytes) is not there yet -- run the chain once before recording it", name, bytes);
            return nullptr;
        }
        if (b.p) {
            (void)hipFree(b.p);
            ++ws_epoch;
        }
        b.p = nullptr;
        b.bytes = 0;
        // small buffers get 12.5 % of slack (a slightly larger request re-uses them); the large ones -- path state, ray and
        // radiance records, sized by the pass -- are allocated as asked
        const size_t want = bytes < (size_t(64) << 20) ? bytes + bytes / 8 + 256 : bytes + 256;
        if (ws_limit && ws_total() + want > ws_limit) {  // make room: what this call has not asked for goes first
            for (auto it = ws.begin(); it != ws.end();) {
                if (it->second.stamp != call_seq && &it->second != &b) {
                    if (it->second.p) {
                        (void)hipFree(it->second.p);
                        ++ws_epoch;
                    }
                    it = ws.erase(it);
                } else {
                    ++it;
                }
            }
        }
        if (ws_limit && ws_total() + want > ws_limit) {
            fail(PBRT_E_NOMEM, "workspace limit: %s wants %zu bytes on top of %zu held, limit %zu", name, want, ws_total(), ws_limit);
            return nullptr;
        }
        // PBRT_DEBUG_ALLOC_FAIL_BYTES (tests of the halve-the-pass retry): a request above this size fails the way a hipMalloc
        // that lost the race against another allocator does
        const char *dbg_fail = getenv("PBRT_DEBUG_ALLOC_FAIL_BYTES");  // read per allocation: a test sets and clears it
        hipError_t e = (dbg_fail && want > (size_t)strtoull(dbg_fail, nullptr, 0)) ? hipErrorOutOfMemory : hipMalloc(&b.p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fail(PBRT_E_NOMEM, "hipMalloc(%zu) for %s: %s", want, name, hipGetErrorString(e));
            b.p = nullptr;
            return nullptr;
        }
        b.bytes = want;
        return b.p;
    }
    hipEvent_t event(size_t i) {
        while (ev_pool.size() <= i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            ev_pool.push_back(e);
        }
        return ev_pool[i];
    }
};

struct pbrt_scene {
    pbrt_ctx *ctx = nullptr;
    DevScene ds{};
    int accel_kernel = ACCEL_K_BRUTE;
    uint32_t lds_bytes = 0;
    uint32_t bvh_depth = 0;  // levels of inner nodes of the BVH4
    bool curved = true;      // the scene holds spheres or cones (else the BVH stream kernels run without their tests)
    std::vector<void *> allocs;
    pbrt_material *d_mats = nullptr;
    uint32_t n_mats = 0;
    // fuse plan learnt from the path survival of the last render of this scene (brute-force kernels; 0: none yet)
    uint32_t plan_hint = 0;
    bool plan_hint_valid = false;
};

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return (ctx)->fail(PBRT_E_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define NEED(ctx, cond)                                                        \
    do {                                                                       \
        if (!(cond)) return (ctx)->fail(PBRT_E_INVALID, "invalid argument: %s", #cond); \
    } while (0)

// entry points that wait, copy from host memory or free: not while the context's stream records (pbrt_ctx_record_begin)
#define NOT_RECORDING(ctx)                                                                                                   \
    do {                                                                                                                     \
        if ((ctx)->recording) return (ctx)->refuse((std::string(__func__) + " cannot run").c_str());                         \
    } while (0)

// finishes an acquisition that was queued without waiting (pbrt_us_acquire_queue_dev): waits for the stream, checks the guard words,
// fills pbrt_stats.  Called first by every entry point that waits for the stream or starts other work on the context.
static int us_finish(pbrt_ctx *c);
static inline int ctx_settle(pbrt_ctx *c) {
    // (every entry point that waits for the stream or starts work of its own comes through here: none of them may run while the
    // stream records -- a wait would invalidate the capture)
    if (c->recording) return c->refuse("only the queueing entry points may be called");
    return c->pend.active ? us_finish(c) : PBRT_OK;
}

template <typename T>
static int upload(pbrt_scene *s, const T *src, size_t n, const T **dst) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(n * sizeof(T), 16);
    HIPCHK(s->ctx, hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    if (n) HIPCHK(s->ctx, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T *>(p);
    return PBRT_OK;
}

static inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Primitives that can occlude a segment between two points of the scene (DevScene::occ_prims).  A planar
// primitive is dropped when all OTHER geometry and every point emitter lie in one closed half-space of its
// plane: it then sits on the boundary of the scene's convex hull and a segment whose end points are in the
// hull cannot cross it.  Exact comparisons (>= 0 in f64 on the f32 data), so nearly-coplanar scenes just
// keep their primitives.
#ifdef PBRT_BRUTE_PAIRS
// Consecutive planar primitives two by two, their v0 / e1 / e2 interleaved; everything else (spheres) on its own.
static std::vector<PairItem> build_pair_items(const pbrt_prim *prims, uint32_t n) {
    auto planar = [](const pbrt_prim &P) { return P.type == PBRT_PRIM_TRIANGLE || P.type == PBRT_PRIM_PARALLELOGRAM; };
    std::vector<PairItem> items;
    for (uint32_t i = 0; i < n;) {
        PairItem it;
        std::memset(&it, 0, sizeof it);
        it.dw[21] = i;
        if (!planar(prims[i])) {
            std::memcpy(it.dw, &prims[i], sizeof(pbrt_prim));
            it.dw[18] = prims[i].type;
            it.dw[19] = 0xffffffffu;
            it.dw[20] = 1;
            it.dw[21] = i;
            i += 1;
        } else {
            const bool two = i + 1 < n && planar(prims[i + 1]);
            const pbrt_prim &A = prims[i], &B = prims[two ? i + 1 : i];
            for (int k = 0; k < 9; ++k) {
                std::memcpy(&it.dw[2 * k], &A.g[k], 4);
                std::memcpy(&it.dw[2 * k + 1], &B.g[k], 4);
            }
            it.dw[18] = A.type;
            it.dw[19] = two ? B.type : 0xffffffffu;
            it.dw[20] = 0;
            i += two ? 2 : 1;
        }
        items.push_back(it);
    }
    return items;
}
#endif

static std::vector<pbrt_prim> find_occluders(const pbrt_scene_desc *d) {
    std::vector<pbrt_prim> occ;
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const pbrt_prim &P = d->prims[i];
        bool keep = true;
        if (P.type == PBRT_PRIM_TRIANGLE || P.type == PBRT_PRIM_PARALLELOGRAM) {
            const double n[3] = {P.g[9], P.g[10], P.g[11]}, p0[3] = {P.g[0], P.g[1], P.g[2]};
            double lo = INFINITY, hi = -INFINITY;
            auto side = [&](double x, double y, double z) {
                return n[0] * (x - p0[0]) + n[1] * (y - p0[1]) + n[2] * (z - p0[2]);
            };
            auto acc = [&](double s) {
                lo = std::min(lo, s);
                hi = std::max(hi, s);
            };
            for (uint32_t j = 0; j < d->n_prims; ++j) {
                if (j == i) continue;
                const pbrt_prim &Q = d->prims[j];
                if (Q.type == PBRT_PRIM_SPHERE) {
                    double s = side(Q.g[0], Q.g[1], Q.g[2]);
                    acc(s - (double)Q.g[3]);
                    acc(s + (double)Q.g[3]);
                } else if (Q.type == PBRT_PRIM_CONE) {  // extent of the base ellipse along n, and the apex
                    double cc[3], ca[3], cb[3], cx[3];
                    cone_world_frame(Q, cc, ca, cb, cx);
                    const double s = side(cc[0], cc[1], cc[2]);
                    const double an = n[0] * ca[0] + n[1] * ca[1] + n[2] * ca[2], bn = n[0] * cb[0] + n[1] * cb[1] + n[2] * cb[2];
                    const double r = std::sqrt(an * an + bn * bn) * (1.0 + 1e-12);
                    acc(s - r);
                    acc(s + r);
                    acc(side(cx[0], cx[1], cx[2]));
                } else {
                    const double v0[3] = {Q.g[0], Q.g[1], Q.g[2]}, e1[3] = {Q.g[3], Q.g[4], Q.g[5]}, e2[3] = {Q.g[6], Q.g[7], Q.g[8]};
                    acc(side(v0[0], v0[1], v0[2]));
                    acc(side(v0[0] + e1[0], v0[1] + e1[1], v0[2] + e1[2]));
                    acc(side(v0[0] + e2[0], v0[1] + e2[1], v0[2] + e2[2]));
                    if (Q.type == PBRT_PRIM_PARALLELOGRAM)
                        acc(side(v0[0] + e1[0] + e2[0], v0[1] + e1[1] + e2[1], v0[2] + e1[2] + e2[2]));
                }
            }
            for (uint32_t e = 0; e < d->n_emitters; ++e)
                if (d->emitters[e].type == PBRT_EMIT_POINT)
                    acc(side(d->emitters[e].pos[0], d->emitters[e].pos[1], d->emitters[e].pos[2]));
            if (lo >= 0.0 || hi <= 0.0) keep = false;
            // a scene lit by ONE single-primitive area light: every shadow segment ends (1 - ShadowEpsilon)
            // short of that primitive's plane, so the light itself never occludes
            if (d->n_emitters == 1 && d->emitters[0].type == PBRT_EMIT_AREA && d->emitters[0].count == 1 &&
                d->light_prims[d->emitters[0].first] == i)
                keep = false;
        }
        if (keep) occ.push_back(P);
    }
    return occ;
}

extern "C" {

int pbrt_abi_version(void) { return PBRT_ABI_VERSION; }

int pbrt_ctx_create(int device, pbrt_ctx **out) {
    if (!out) {
        g_ctxless_error = "pbrt_ctx_create: out is NULL";
        return PBRT_E_INVALID;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        g_ctxless_error = std::string("no HIP device visible: ") + hipGetErrorString(e) +
                          " (the ray-transport hot path has no CPU fallback)";
        return PBRT_E_DEVICE;
    }
    if (device < 0 || device >= n) {
        g_ctxless_error = "device index out of range";
        return PBRT_E_INVALID;
    }
    pbrt_ctx *c = new pbrt_ctx();
    c->device = device;
#ifdef PBRT_CU_MASK_PROBE  // diagnostic builds: run on a subset of the CUs (PBRT_CU_MASK = even | odd | low | pairs)
    auto masked_stream = [&](hipStream_t *st) -> hipError_t {
        const char *m = getenv("PBRT_CU_MASK");
        if (!m) return hipStreamCreate(st);
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t b = 0; b < 256; ++b) {
            bool on = !strcmp(m, "even") ? (b & 1u) == 0 : !strcmp(m, "odd") ? (b & 1u) == 1 : !strcmp(m, "low") ? b < 128
                      : !strcmp(m, "altcu") ? ((b >> 5) & 1u) == 0 : !strcmp(m, "altse") ? ((b >> 3) & 1u) == 0 : !strcmp(m, "altcu2") ? ((b >> 6) & 1u) == 0 : !strcmp(m, "pairs") ? (b & 2u) == 0 : true;
            if (on) mask[b >> 5] |= 1u << (b & 31u);
        }
        return hipExtStreamCreateWithCUMask(st, 8, mask);
    };
    if ((e = hipSetDevice(device)) != hipSuccess || (e = masked_stream(&c->stream)) != hipSuccess ||
#else
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&c->stream)) != hipSuccess ||
#endif
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        g_ctxless_error = std::string("device init: ") + hipGetErrorString(e);
        delete c;
        return PBRT_E_DEVICE;
    }
    if ((e = hipHostMalloc(&c->pinned, PIN_BYTES, hipHostMallocDefault)) != hipSuccess) {
        g_ctxless_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        (void)hipStreamDestroy(c->stream);
        delete c;
        return PBRT_E_NOMEM;
    }
    std::memset(c->pinned, 0, PIN_BYTES);
    if (const char *lim = getenv("PBRT_WORKSPACE_LIMIT_BYTES")) c->ws_limit = (size_t)strtoull(lim, nullptr, 0);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->lds_limit = (uint32_t)prop.sharedMemPerBlock;
        if (prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
    }
    *out = c;
    return PBRT_OK;
}

int pbrt_ctx_destroy(pbrt_ctx *c) {
    if (!c) return PBRT_OK;
    (void)hipSetDevice(c->device);
    if (c->recording) {  // an open recording: close the capture and drop it
        hipGraph_t g = nullptr;
        if (hipStreamEndCapture(c->stream, &g) == hipSuccess && g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        c->recording = false;
        c->pend.active = false;
    }
    (void)ctx_settle(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->st_trace) (void)hipStreamSynchronize(c->st_trace);
    if (c->st_shade) (void)hipStreamSynchronize(c->st_shade);
    for (auto &kv : c->ws)
        if (kv.second.p) (void)hipFree(kv.second.p);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto e : c->sync_ev) (void)hipEventDestroy(e);
    for (auto e : c->img_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->st_trace) (void)hipStreamDestroy(c->st_trace);
    if (c->st_shade) (void)hipStreamDestroy(c->st_shade);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PBRT_OK;
}

const char *pbrt_last_error(pbrt_ctx *c) { return c ? c->err.c_str() : g_ctxless_error.c_str(); }

int pbrt_ctx_set_workspace_limit(pbrt_ctx *c, uint64_t bytes) {
    if (!c) return PBRT_E_INVALID;
    if (int rc = ctx_settle(c)) return rc;
    c->ws_limit = (size_t)bytes;
    if (bytes && c->ws_total() > bytes) {  // a limit below what the context holds: everything goes back (calls are synchronous, nothing is in use)
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (auto &kv : c->ws)
            if (kv.second.p) (void)hipFree(kv.second.p);
        c->ws.clear();
        c->stats.workspace_bytes = 0;
    }
    return PBRT_OK;
}

// frees every workspace buffer the most recent call did not use, and every one that is larger than that call needed
int pbrt_ctx_trim(pbrt_ctx *c, uint64_t *held_after) {
    if (!c) return PBRT_E_INVALID;
    if (int rc = ctx_settle(c)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto it = c->ws.begin(); it != c->ws.end();) {
        DevBuf &b = it->second;
        const size_t fit = b.need < (size_t(64) << 20) ? b.need + b.need / 8 + 256 : b.need + 256;
        if (b.stamp != c->call_seq || b.bytes > fit || !b.p) {
            if (b.p) {
                (void)hipFree(b.p);
                ++c->ws_epoch;
            }
            it = c->ws.erase(it);
        } else {
            ++it;
        }
    }
    c->stats.workspace_bytes = c->ws_total();
    if (held_after) *held_after = c->stats.workspace_bytes;
    return PBRT_OK;
}

int pbrt_get_stats(pbrt_ctx *c, pbrt_stats *out) {
    if (!c || !out) return PBRT_E_INVALID;
    if (int rc = ctx_settle(c)) return rc;  // (a queued acquisition: its counters arrive now)
    *out = c->stats;
    return PBRT_OK;
}

int pbrt_scene_create(pbrt_ctx *c, const pbrt_scene_desc *d, pbrt_scene **out) {
    if (!c) return PBRT_E_INVALID;
    NEED(c, d && out);
    NOT_RECORDING(c);
    NEED(c, d->n_prims > 0 && d->prims && d->n_materials > 0 && d->materials);
    NEED(c, d->n_emitters == 0 || d->emitters);
    NEED(c, d->n_light_prims == 0 || (d->light_prims && d->light_cdf));
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const pbrt_prim &p = d->prims[i];
        if (p.type > PBRT_PRIM_CONE) return c->fail(PBRT_E_UNSUPPORTED, "primitive %u: type %u is not supported", i, p.type);
        if (p.type == PBRT_PRIM_CONE) {
            double cc[3], ca[3], cb[3], cx[3];
            if (!cone_world_frame(p, cc, ca, cb, cx))
                return c->fail(PBRT_E_INVALID, "primitive %u: cone needs an invertible, finite world -> object matrix", i);
        }
        if (p.material >= d->n_materials) return c->fail(PBRT_E_INVALID, "primitive %u: material out of range", i);
        if (p.emitter >= 0 && (uint32_t)p.emitter >= d->n_emitters)
            return c->fail(PBRT_E_INVALID, "primitive %u: emitter out of range", i);
    }
    for (uint32_t i = 0; i < d->n_emitters; ++i) {
        const pbrt_emitter &e = d->emitters[i];
        if (e.type == PBRT_EMIT_AREA) {
            if (e.count == 0 || e.first + e.count > d->n_light_prims)
                return c->fail(PBRT_E_INVALID, "emitter %u: light primitive range out of bounds", i);
            for (uint32_t k = 0; k < e.count; ++k) {
                uint32_t pi = d->light_prims[e.first + k];
                if (pi >= d->n_prims || d->prims[pi].type == PBRT_PRIM_SPHERE || d->prims[pi].type == PBRT_PRIM_CONE)
                    return c->fail(PBRT_E_UNSUPPORTED, "emitter %u: area lights need triangle/parallelogram primitives", i);
            }
        } else if (e.type != PBRT_EMIT_POINT) {
            return c->fail(PBRT_E_INVALID, "emitter %u: unknown type", i);
        }
    }
    if (d->vertex_normals)
        for (size_t i = 0; i < (size_t)d->n_prims * 9; ++i)
            if (!std::isfinite(d->vertex_normals[i])) return c->fail(PBRT_E_INVALID, "vertex_normals: entry %zu is not finite", i);
    HIPCHK(c, hipSetDevice(c->device));
    pbrt_scene *s = new pbrt_scene();
    s->ctx = c;
    int rc;
    const pbrt_prim *d_prims_by_id = nullptr;
#define UP(call)               \
    if ((rc = (call)) != 0) {  \
        pbrt_scene_destroy(s); \
        return rc;             \
    }
    UP(upload(s, d->prims, d->n_prims, &d_prims_by_id));
    s->ds.prims_by_id = d_prims_by_id;
    s->ds.n_prims = d->n_prims;
    const pbrt_material *dm = nullptr;
    UP(upload(s, d->materials, d->n_materials, &dm));
    s->ds.mats = dm;
    s->d_mats = const_cast<pbrt_material *>(dm);
    s->n_mats = s->ds.n_mats = d->n_materials;
    UP(upload(s, d->emitters, d->n_emitters, &s->ds.emitters));
    s->ds.n_emitters = d->n_emitters;
    UP(upload(s, d->light_prims, d->n_light_prims, &s->ds.light_prims));
    UP(upload(s, d->light_cdf, d->n_light_prims, &s->ds.light_cdf));
    s->ds.n_light_prims = d->n_light_prims;
    NEED(c, d->accel <= PBRT_ACCEL_BVH_GLOBAL);
    const bool want_bvh = d->accel == PBRT_ACCEL_BVH || d->accel == PBRT_ACCEL_BVH_GLOBAL ||
                          (d->accel == PBRT_ACCEL_AUTO && d->n_prims > 32);
    if (!want_bvh) {
        s->ds.prims = d_prims_by_id;
        s->ds.lprims = nullptr;
        s->ds.nodes = nullptr;
        s->ds.n_nodes = 0;
        bool small = d->n_prims <= TAB_MAX && d->n_materials <= TAB_MAX && d->n_emitters <= TAB_MAX;
        for (uint32_t i = 0; i < d->n_prims; ++i)
            if (d->prims[i].type == PBRT_PRIM_CONE) small = false;  // only the _BIG variant carries the cone code
        if (d->vertex_normals) small = false;                      // ... and the shading-normal code
        if (d->vertex_normals) UP(upload(s, d->vertex_normals, (size_t)d->n_prims * 9, &s->ds.vnormals));
        s->accel_kernel = small ? ACCEL_K_BRUTE : ACCEL_K_BRUTE_BIG;
        std::vector<pbrt_prim> occ = find_occluders(d);
        UP(upload(s, occ.data(), occ.size(), &s->ds.occ_prims));
        s->ds.n_occ = (uint32_t)occ.size();
#ifdef PBRT_BRUTE_PAIRS
        if (s->accel_kernel == ACCEL_K_BRUTE) {  // pair records for brute_closest_pairs (device_scene.h)
            std::vector<PairItem> items = build_pair_items(d->prims, d->n_prims);
            UP(upload(s, items.data(), items.size(), &s->ds.pair_items));
            s->ds.n_pair_items = (uint32_t)items.size();
        }
#endif
    } else {
        HostBvh bvh;
        // (a tree whose leaf records alone exceed the LDS stays in global memory whatever its shape: the SAH constant of those trees)
        const bool sure_global = !c->lds_limit || (size_t)d->n_prims * sizeof(DevLeafPrim) > c->lds_limit || d->accel == PBRT_ACCEL_BVH_GLOBAL;
        build_bvh(d->prims, d->n_prims, &bvh, sure_global ? BVH_CTRAV_GLOBAL : BVH_CTRAV);
        HostBvh4 b4;
        to_bvh4(bvh, &b4);
        // traversal stack: one entry per level of inner nodes, 24-bit node indices (device_scene.h BvhStack)
        if (b4.depth > BVH_STK_MAX || b4.nodes.size() >= (1u << 24) - 1u || d->n_prims >= (1u << 27)) {
            pbrt_scene_destroy(s);
            return c->fail(PBRT_E_UNSUPPORTED, "BVH depth %u / %zu nodes exceed the traversal state", b4.depth, b4.nodes.size());
        }
        std::vector<HostLeafPrim> lp;
        make_leaf_prims(d->prims, bvh.order, &lp);
        static_assert(sizeof(HostLeafPrim) == sizeof(DevLeafPrim) && sizeof(DevLeafPrim) == 40, "leaf record layout");
        static_assert(sizeof(HostNode4) == sizeof(DevNode4) && sizeof(DevNode4) == 64, "node layout");
        s->ds.prims = d_prims_by_id;  // Hit::slot is the caller's index
        const DevLeafPrim *dl = nullptr;
        UP(upload(s, reinterpret_cast<const DevLeafPrim *>(lp.data()), lp.size(), &dl));
        s->ds.lprims = dl;
        if (d->vertex_normals) UP(upload(s, d->vertex_normals, (size_t)d->n_prims * 9, &s->ds.vnormals));
        const DevNode4 *dn = nullptr;
        UP(upload(s, reinterpret_cast<const DevNode4 *>(b4.nodes.data()), b4.nodes.size(), &dn));
        s->ds.nodes = dn;
        s->ds.n_nodes = (uint32_t)b4.nodes.size();
        s->bvh_depth = b4.depth;
        s->curved = false;
        for (uint32_t i = 0; i < d->n_prims; ++i)
            if (d->prims[i].type != PBRT_PRIM_TRIANGLE && d->prims[i].type != PBRT_PRIM_PARALLELOGRAM) s->curved = true;
        // the LDS image: 56 bytes per node (planes, device_scene.h TreeLds) + the leaf records
        const size_t lds = b4.nodes.size() * LDS_IMAGE_NODE_BYTES + (size_t)d->n_prims * sizeof(DevLeafPrim);
        // beside the image: the traversal stacks of a 1024-thread workgroup and the kernels' small static arrays (1 KiB of slack)
        if (c->lds_limit && lds + BVH_STK_DW(SEG_BVH) * 4 + 1024 <= c->lds_limit && d->accel != PBRT_ACCEL_BVH_GLOBAL) {
            s->accel_kernel = ACCEL_K_BVH_LDS;
            s->lds_bytes = (uint32_t)((lds + 15) & ~size_t(15));
        } else {
            s->accel_kernel = ACCEL_K_BVH_GLOBAL;
        }
    }
#undef UP
    *out = s;
    return PBRT_OK;
}

int pbrt_scene_update_material(pbrt_scene *s, uint32_t index, const pbrt_material *m) {
    if (!s) return PBRT_E_INVALID;
    pbrt_ctx *c = s->ctx;
    NEED(c, m && index < s->n_mats);
    NOT_RECORDING(c);  // (a recorded copy would replay the bytes of THIS call's host block)
    HIPCHK(c, hipSetDevice(c->device));
    // in the order of the context's stream: behind an acquisition that is still queued, ahead of the next one (the 32 bytes are
    // staged before the call returns)
    HIPCHK(c, hipMemcpyAsync(s->d_mats + index, m, sizeof *m, hipMemcpyHostToDevice, c->stream));
    return PBRT_OK;
}

int pbrt_scene_destroy(pbrt_scene *s) {
    if (!s) return PBRT_OK;
    NOT_RECORDING(s->ctx);
    (void)hipSetDevice(s->ctx->device);
    (void)ctx_settle(s->ctx);
    (void)hipStreamSynchronize(s->ctx->stream);  // queued work may still read the scene
    ++s->ctx->ws_epoch;
    for (void *p : s->allocs) (void)hipFree(p);
    delete s;
    return PBRT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// radiance mode driver
// ------------------------------------------------------------------------------------------------
// Depths at which a launch of the brute-force kernels walks two bounces (bit d: bounces d and d + 1).  Measured on the
// Cornell box (0x1 while the two-bounce kernels ran at 6 - 7 waves per SIMD, every pair since they run
// at 8); pbrt_film_desc.flags can override it per call (PBRT_FILM_FUSE_PLAN).
#ifndef PBRT_DEFAULT_FUSE_PLAN
#define PBRT_DEFAULT_FUSE_PLAN 0x15u
#endif
// Depth from which one launch walks every remaining bounce of a pass (k_walk; 0xff: never).  PBRT_FILM_WALK_FROM overrides.
#ifndef PBRT_DEFAULT_WALK_FROM
#define PBRT_DEFAULT_WALK_FROM 0xffu
#endif

#ifdef PBRT_DIAG
template <bool FIRST>
static void launch_walk(pbrt_scene *s, const RadArgs &a, uint32_t nseg, uint32_t nb0) {
    hipStream_t st = s->ctx->stream;
    if (s->accel_kernel == ACCEL_K_BRUTE) {
        if (nb0 == 2)
            hipLaunchKernelGGL((k_walk<FIRST, ACCEL_K_BRUTE, 2>), dim3(nseg), dim3(SEG_BRUTE), 0, st, a);
        else
            hipLaunchKernelGGL((k_walk<FIRST, ACCEL_K_BRUTE, 1>), dim3(nseg), dim3(SEG_BRUTE), 0, st, a);
    } else {
        if (nb0 == 2)
            hipLaunchKernelGGL((k_walk<FIRST, ACCEL_K_BRUTE_BIG, 2>), dim3(nseg), dim3(SEG_BRUTE), 0, st, a);
        else
            hipLaunchKernelGGL((k_walk<FIRST, ACCEL_K_BRUTE_BIG, 1>), dim3(nseg), dim3(SEG_BRUTE), 0, st, a);
    }
}
#endif
// In-kernel repack of a chain launch (kernels_radiance.h k_bounce, REPACK): after which of its bounces the workgroup packs its live
// paths together.  PBRT_CHAIN_REPACK=mask overrides (0: never).
static uint32_t chain_repack_mask() {
    static const char *e = getenv("PBRT_CHAIN_REPACK");
    return e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
}
// nb: bounces this launch walks (>= 2: the multi-bounce variants of the brute-force kernels, kernels_radiance.h; a.nb = nb)
template <bool FIRST>
static int launch_bounce(pbrt_scene *s, const RadArgs &a, uint32_t nseg, uint32_t nb = 1) {
    hipStream_t st = s->ctx->stream;
    switch (s->accel_kernel) {
        case ACCEL_K_BRUTE:
            if (nb >= 2)
                hipLaunchKernelGGL((k_bounce<FIRST, ACCEL_K_BRUTE, 2>), dim3(nseg), dim3(SEG_BRUTE), 0, st, a);
            else
                hipLaunchKernelGGL((k_bounce<FIRST, ACCEL_K_BRUTE>), dim3(nseg), dim3(SEG_BRUTE), 0, st, a);
            break;
        case ACCEL_K_BRUTE_BIG:
            if (nb >= 2)
                hipLaunchKernelGGL((k_bounce<FIRST, ACCEL_K_BRUTE_BIG, 2>), dim3(nseg), dim3(SEG_BRUTE), 0, st, a);
            else
                hipLaunchKernelGGL((k_bounce<FIRST, ACCEL_K_BRUTE_BIG>), dim3(nseg), dim3(SEG_BRUTE), 0, st, a);
            break;
#ifdef PBRT_DIAG  // the fused BVH bounce (PBRT_FILM_NO_HIT_POOL): diagnostic build only
        case ACCEL_K_BVH_GLOBAL:
            hipLaunchKernelGGL((k_bounce<FIRST, ACCEL_K_BVH_GLOBAL>), dim3(nseg), dim3(SEG_BVH), 0, st, a);
            break;
        default:
            hipLaunchKernelGGL((k_bounce<FIRST, ACCEL_K_BVH_LDS>), dim3(nseg), dim3(SEG_BVH), s->lds_bytes, st, a);
            break;
#else
        default:  // BVH scenes run k_trace / k_shade (wf_bounces); the fused BVH bounce exists in the diagnostic build only
            return s->ctx->fail(PBRT_E_UNSUPPORTED, "launch_bounce: no fused bounce kernel for accelerator %d in this build", s->accel_kernel);
#endif
    }
    return PBRT_OK;
}

static int set_lds_attr(pbrt_scene *s) {
    if (s->accel_kernel != ACCEL_K_BVH_LDS) return PBRT_OK;
    pbrt_ctx *c = s->ctx;
    int bytes = (int)s->lds_bytes;
#ifdef PBRT_DIAG
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bounce<true, ACCEL_K_BVH_LDS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bounce<false, ACCEL_K_BVH_LDS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_us_bounce<true, ACCEL_K_BVH_LDS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_us_bounce<false, ACCEL_K_BVH_LDS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
#else
    (void)c;
    (void)bytes;
#endif
    return PBRT_OK;
}

// Fuse plan: bit d set = the launch that walks bounce d goes on with bounce d + 1 in registers (at most MAX_CHAIN bounces per
// launch).  0 = one launch per bounce, 0x15 = pairs, all ones = as many bounces per launch as MAX_CHAIN allows.
static uint32_t chain_len(uint32_t plan, uint32_t depth, uint32_t max_depth) {
    uint32_t nb = 1;
    while (nb < MAX_CHAIN && depth + nb < max_depth && depth + nb - 1 < 32 && ((plan >> (depth + nb - 1)) & 1u)) ++nb;
    return nb;
}
// The plan that pays for a scene follows from how many paths survive each bounce: walking bounce d + 1 in the same launch saves
// the state round trip of the survivors and costs the idle lanes of the others.  Measured (cbox, survival 0.87 / 0.77 / 0.83 /
// 0.85 / 0.20: ONE launch for all six bounces 6.23 ms, triples 6.30, pairs 6.58; open scenes with survival 0.37 - 0.49 / 0.29 /
// 0.33: pairs 2.11 / 4.52 ms, triples 2.11 / 4.84, one launch 2.47 / 5.70): go on while at least 45 % of the paths do; the last
// bounce of a path only looks for emitters and is always taken along.  Any plan renders the same film.
static uint32_t plan_from_survival(const unsigned long long *live, uint32_t max_depth) {
    uint32_t plan = 0;
    for (uint32_t d = 0; d + 1 < max_depth && d < 31; ++d) {
        const bool last = d + 2 == max_depth;
        if (live[d] == 0) break;
        if (last || (double)live[d + 1] >= 0.45 * (double)live[d]) plan |= 1u << d;
    }
    return plan;
}

// k_chain_pair (two tiles per wave, kernels_radiance.h; lost its A/B): diagnostic builds launch it instead of the k_bounce chain
// when PBRT_PAIR_MERGE=k names the merge bounce and the plan walks the whole path in one launch.
#ifdef PBRT_DIAG
static uint32_t pair_merge_bounce(uint32_t plan, uint32_t max_depth) {
    const char *e = getenv("PBRT_PAIR_MERGE");  // (read per pass: a test switches it between renders)
    if (!e || max_depth > MAX_CHAIN || max_depth < 2 || chain_len(plan, 0, max_depth) < max_depth) return 0;
    const uint32_t k = (uint32_t)atoi(e);
    return k < max_depth ? k : 0u;
}
#endif

// Byte model of the radiance path (DESIGN.md "Algorithmic bytes").  live[d] = paths entering depth d.
// A launch that walks two bounces (fuse plan bit d) keeps its paths in registers between them: the survivors of bounce d
// are neither written nor read back, only the survivors of bounce d + 1 are.
// hits (k_bounce_pool launches, BVH scenes, bounces >= 1; else nullptr): hits[d] = rays of depth d that hit something.  Every ray
// reads its origin and direction (24 B), a path that hit something reads its full state.
static void radiance_model_bytes(const unsigned long long *live, uint32_t nd, uint64_t samples, uint64_t film_px,
                                 uint32_t passes, uint32_t fuse_plan, uint32_t max_depth, const unsigned long long *hits,
                                 uint64_t *total, uint64_t *bounce) {
    uint64_t b = 0;
    for (uint32_t d = 0; d < nd;) {
        const uint32_t nb = std::min(chain_len(fuse_plan, d, max_depth), nd - d);
        uint64_t in = live[d], next = (d + nb < nd) ? live[d + nb] : 0;
        if (d > 0 && hits)
            b += in * 24 + hits[d] * (N_STATE * 4);
        else if (d > 0)
            b += in * (N_STATE * 4);  // state read
        b += next * (N_STATE * 4);            // compacted survivors written
        b += (in - next) * 12;                // radiance of the paths that ended (at either bounce of the launch)
        d += nb;
    }
    *bounce = b;
    *total = b + samples * 12 /* film gather reads Lhome once */ + film_px * 32ull * passes /* accumulator RMW */ +
             film_px * 28 /* resolve */;
}


// ---- BVH scenes: intersection and shading as separate streams (kernels_wavefront.h) -------------------------------------------
struct WfPlan {
    bool packet = true;      // camera rays: one tree walk per 64-path tile (k_trace_primary); PBRT_WF_PACKET=0: k_trace<true> (A/B)
    uint32_t grid_deep = 2;  // workgroups per CU from bounce 2 on (few rays: a resident round of larger shares; ring 8 / 2 / 1: 139.7 / 137.1 / 134.9 ms)
    uint32_t threads = 1024, rows = 2, grid_mult = 8;  // grid: ring 1024^2 x 64: 2 / 4 / 8 / 16 workgroups per CU -> 27.6 / 21.3 / 20.4 / 21.3 ms
    size_t lds = 0;
};
// Workgroup shape of k_trace: the image plus (rows + 1) stack rows per workgroup; two 1024-thread workgroups per CU when both fit
// (8 waves per SIMD at <= 64 VGPRs), else one.  PBRT_WF_THREADS / PBRT_WF_ROWS / PBRT_WF_GRID_MULT override (diagnostic A/B).
static WfPlan wf_plan(const pbrt_scene *s) {
    WfPlan p;
    static const char *e_thr = getenv("PBRT_WF_THREADS"), *e_rows = getenv("PBRT_WF_ROWS"), *e_grid = getenv("PBRT_WF_GRID_MULT");
    const uint32_t limit = s->ctx->lds_limit ? s->ctx->lds_limit : 65536u;
    const uint32_t image = s->accel_kernel == ACCEL_K_BVH_LDS ? s->lds_bytes : 0u;
    // 1024-thread workgroups for trees in global memory as well: what counts is the size of the queue a workgroup's waves share
    // (bunny.ply 1024^2 x 64: 256 / 512 / 1024 threads 38.5 / 31.6 / 30.2 ms), not the LDS -- a copy of the top of the tree (the
    // first 16 .. 1008 nodes in breadth-first order) in LDS on top of that was worth 1 - 2 %: the vector caches hold those nodes anyway
    p.threads = 1024u;
    if (e_thr) {  // a power of two (the stack rows are addressed by a shift)
        const uint32_t want = (uint32_t)atoi(e_thr);
        p.threads = want >= 1024u ? 1024u : want >= 512u ? 512u : want >= 256u ? 256u : want >= 128u ? 128u : 64u;
    }
    const uint32_t statics = 1024;  // queue words and the segment table, with slack
    uint32_t budget = limit / 2;     // two workgroups per CU
    if (image + statics + 3u * p.threads * 4u > budget) budget = limit;
    const uint32_t rows_fit = (budget - image - statics) / (p.threads * 4u);
    p.rows = std::max(2u, std::min(rows_fit, 8u));
    if (e_rows) p.rows = std::max(2u, std::min((uint32_t)atoi(e_rows), 15u));
    if (e_grid) p.grid_mult = std::max(1u, (uint32_t)atoi(e_grid));
    static const char *e_deep = getenv("PBRT_WF_GRID_DEEP");
    if (e_deep) p.grid_deep = std::max(1u, (uint32_t)atoi(e_deep));
    static const char *e_pkt = getenv("PBRT_WF_PACKET");
    if (e_pkt) p.packet = atoi(e_pkt) != 0;
    if (3u * s->bvh_depth > 64u) p.packet = false;  // the wave's stack is the 64 lanes of one register (bvh_packet_closest)
    p.lds = (size_t)image + (size_t)p.rows * p.threads * 4u;
    return p;
}
static int wf_set_attr(pbrt_scene *s, const WfPlan &p) {
    pbrt_ctx *c = s->ctx;
    if (s->accel_kernel == ACCEL_K_BVH_LDS) {
#define WF_ATTR(fn, bytes) HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
        if (s->curved) {
            WF_ATTR((k_trace<true, ACCEL_K_BVH_LDS, true>), p.lds);
            WF_ATTR((k_trace<false, ACCEL_K_BVH_LDS, true>), p.lds);
            WF_ATTR((k_trace_primary<ACCEL_K_BVH_LDS, true>), s->lds_bytes);
        } else {
            WF_ATTR((k_trace<true, ACCEL_K_BVH_LDS, false>), p.lds);
            WF_ATTR((k_trace<false, ACCEL_K_BVH_LDS, false>), p.lds);
            WF_ATTR((k_trace_primary<ACCEL_K_BVH_LDS, false>), s->lds_bytes);
        }
#undef WF_ATTR
    }
    return PBRT_OK;
}
// the context's guard words (k_trace's turn guard): allocated and cleared once, cleared again after a trip
static uint32_t *wf_guard(pbrt_ctx *c) {
    const bool fresh = c->ws["wf_guard"].p == nullptr;
    uint32_t *g = (uint32_t *)c->buf("wf_guard", WF_GUARD_WORDS * 4);
    if (g && fresh && hipMemsetAsync(g, 0, WF_GUARD_WORDS * 4, c->stream) != hipSuccess) return nullptr;
    return g;
}
#define WF_BYTES_PER_PATH (2 * WF_STATE_Q * 16 + 4 + 2 * 64 + 16)  // two state sets, hit index, two shadow sets, Lhome
// the buffers whose size follows the pass: given back before a retry with half the paths in flight (render_impl, us_impl)
static void release_pass_buffers(pbrt_ctx *c) {
    for (const char *nm : {"wf_stateA", "wf_stateB", "wf_shadowA", "wf_shadowB", "wf_hit_id", "Lhome", "stateA", "stateB"}) c->release(nm);
}
struct WfBufs {
    float4 *stA, *stB, *shA, *shB;
    uint32_t *hit_id, *segA, *segB, *nshA, *nshB;
};
static bool wf_alloc(pbrt_ctx *c, uint32_t cap, uint32_t nreg, WfBufs *b) {
    b->stA = (float4 *)c->buf("wf_stateA", (size_t)cap * WF_STATE_Q * 16);
    b->stB = (float4 *)c->buf("wf_stateB", (size_t)cap * WF_STATE_Q * 16);
    b->hit_id = (uint32_t *)c->buf("wf_hit_id", (size_t)cap * 4);
    b->shA = (float4 *)c->buf("wf_shadowA", (size_t)cap * 64);
    b->shB = (float4 *)c->buf("wf_shadowB", (size_t)cap * 64);
    b->segA = (uint32_t *)c->buf("wf_segA", (size_t)nreg * 4);
    b->segB = (uint32_t *)c->buf("wf_segB", (size_t)nreg * 4);
    b->nshA = (uint32_t *)c->buf("wf_nshA", (size_t)nreg * 4);
    b->nshB = (uint32_t *)c->buf("wf_nshB", (size_t)nreg * 4);
    return b->stA && b->stB && b->hit_id && b->shA && b->shB && b->segA && b->segB && b->nshA && b->nshB;
}
// BVH scenes, opt-in (PBRT_WF_SPLIT=s[,parts]): the CUs are split between the two kernels of a bounce.  k_trace is bound by
// instruction issue and k_shade by HBM, but run side by side on the same CUs they only trade wave slots (round 3: +2.8 %).  Here
// two streams carry CU masks -- s CUs of every shader engine (32 s of the 256) run k_shade, the others k_trace -- and a pass is
// cut into `parts` sets of regions that go through the two streams one phase apart: trace(part, d) -> shade(part, d) ->
// trace(part, d + 1), with trace of one part running beside shade of another.  Regions share nothing (their own slots of every
// buffer, their own statistics rows), so any order renders the same film.
struct WfSplit {
    uint32_t s = 0, parts = 1;  // s = 0: off
};
static WfSplit wf_split_env() {
    WfSplit w;
    static const char *e = getenv("PBRT_WF_SPLIT");
    if (e) {
        unsigned s_ = 0, p_ = 2;
        if (sscanf(e, "%u%*[,:]%u", &s_, &p_) >= 1 && s_ >= 1 && s_ <= 7) {
            w.s = s_;
            w.parts = std::max(2u, std::min(p_, 8u));
        }
    }
    return w;
}
static int wf_split_streams(pbrt_ctx *c, uint32_t s_cus) {
    if (c->st_trace && c->split_s == s_cus) return PBRT_OK;
    if (c->st_trace) {
        (void)hipStreamDestroy(c->st_trace);
        (void)hipStreamDestroy(c->st_shade);
        c->st_trace = c->st_shade = nullptr;
    }
    // bit b of a CU mask: XCC b % 8, shader engine (b >> 3) % 4, CU b >> 5 (tools/cu_mask_probe.hip): the s highest CUs of every
    // shader engine of every XCD shade, so both kernels have CUs on every XCD (their L2s) and every shader engine
    uint32_t mt[8] = {0}, ms[8] = {0};
    for (uint32_t bit = 0; bit < 256; ++bit) ((bit >> 5) >= 8u - s_cus ? ms : mt)[bit >> 5] |= 1u << (bit & 31u);
    HIPCHK(c, hipExtStreamCreateWithCUMask(&c->st_trace, 8, mt));
    HIPCHK(c, hipExtStreamCreateWithCUMask(&c->st_shade, 8, ms));
    c->split_s = s_cus;
    return PBRT_OK;
}

// one k_trace / k_trace_primary launch over a.n_regions regions.  template arguments: <first bounce,> tree in LDS / in global
// memory, scene with curved primitives
static void wf_launch_trace(pbrt_scene *s, const WfArgs &a, const WfPlan &p, uint32_t G, bool first, hipStream_t st) {
    const bool lds = s->accel_kernel == ACCEL_K_BVH_LDS;
#define WF_TRACE(F, A, C) hipLaunchKernelGGL((k_trace<F, A, C>), dim3(G), dim3(p.threads), p.lds, st, a)
#define WF_PRIMARY(A, C) hipLaunchKernelGGL((k_trace_primary<A, C>), dim3(G), dim3(1024), lds ? s->lds_bytes : 0u, st, a)
    const int variant = (lds ? 2 : 0) | (s->curved ? 1 : 0);
    if (first && p.packet) {
        switch (variant) {
            case 3: WF_PRIMARY(ACCEL_K_BVH_LDS, true); break;
            case 2: WF_PRIMARY(ACCEL_K_BVH_LDS, false); break;
            case 1: WF_PRIMARY(ACCEL_K_BVH_GLOBAL, true); break;
            default: WF_PRIMARY(ACCEL_K_BVH_GLOBAL, false); break;
        }
    } else if (first) {
        switch (variant) {
            case 3: WF_TRACE(true, ACCEL_K_BVH_LDS, true); break;
            case 2: WF_TRACE(true, ACCEL_K_BVH_LDS, false); break;
            case 1: WF_TRACE(true, ACCEL_K_BVH_GLOBAL, true); break;
            default: WF_TRACE(true, ACCEL_K_BVH_GLOBAL, false); break;
        }
    } else {
        switch (variant) {
            case 3: WF_TRACE(false, ACCEL_K_BVH_LDS, true); break;
            case 2: WF_TRACE(false, ACCEL_K_BVH_LDS, false); break;
            case 1: WF_TRACE(false, ACCEL_K_BVH_GLOBAL, true); break;
            default: WF_TRACE(false, ACCEL_K_BVH_GLOBAL, false); break;
        }
    }
#undef WF_TRACE
#undef WF_PRIMARY
}
// workgroups of a k_trace launch over nr regions: at least nr / WF_KMAX (a workgroup walks at most WF_KMAX regions), else `mult` per CU
static uint32_t wf_trace_grid(uint32_t nr, uint32_t mult, uint32_t cus) {
    return std::min(nr, std::max(div_up(nr, WF_KMAX), std::max(1u, mult * cus)));
}

// The bounces of one pass.  camera: depth 0 generates its rays from the film keys (else the rays are in b.stA / b.segA).
// Returns the number of launches through *launches.
static int wf_bounces(pbrt_scene *s, WfArgs a, const WfBufs &b, const WfPlan &p, uint32_t nreg, bool camera, uint32_t *launches) {
    pbrt_ctx *c = s->ctx;
    const bool lds = s->accel_kernel == ACCEL_K_BVH_LDS;
    a.lds_bytes = lds ? s->lds_bytes : 0u;
    a.stk_rows = p.rows;
    a.stk_shift = 0;
    while ((1u << a.stk_shift) < p.threads) ++a.stk_shift;
    if (!(a.guard = wf_guard(c))) return PBRT_E_NOMEM;
    a.vis_q = 4;
    // the small shading tables in LDS when they fit (kernels_wavefront.h wf_tables_lds)
    const bool tabs = s->ds.n_mats <= TAB_MAX && s->ds.n_emitters <= TAB_MAX && s->ds.n_light_prims <= TAB_MAX;
    const WfSplit sp = wf_split_env();
    const bool split = sp.s != 0 && c->n_cu == 256 && nreg >= 64u * sp.parts && a.max_depth <= 32;
    const uint32_t parts = split ? sp.parts : 1u;
    if (split) {
        int rc = wf_split_streams(c, sp.s);
        if (rc) return rc;
    }
    const uint32_t cu_trace = split ? 256u - 32u * sp.s : (uint32_t)c->n_cu;
    hipStream_t st_t = split ? c->st_trace : c->stream, st_s = split ? c->st_shade : c->stream;
    uint32_t reg0[8], regn[8];
    for (uint32_t h = 0; h < parts; ++h) {
        reg0[h] = (uint32_t)((uint64_t)nreg * h / parts);
        regn[h] = (uint32_t)((uint64_t)nreg * (h + 1) / parts) - reg0[h];
    }
    // events of the split pipeline: [part][0] = its last k_trace, [part][1] = its last k_shade (a wait captures the record that
    // precedes it, so one event per part and kind serves every depth)
    if (split)
        while (c->sync_ev.size() < 2u * parts + 1u) {
            hipEvent_t e;
            HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->sync_ev.push_back(e);
        }
    float4 *in = b.stA, *out = b.stB, *shi = b.shA, *sho = b.shB;
    uint32_t *sin = b.segA, *sout = b.segB, *ni = b.nshA, *no = b.nshB;
    auto fill = [&](uint32_t depth, bool have_shadows, uint32_t h) {
        a.depth = depth;
        a.st_in = in;
        a.st_out = out;
        a.hit_id = b.hit_id;
        a.shd_in = shi;
        a.shd_out = sho;
        a.seg_in = sin;
        a.seg_out = sout;
        a.nsh_in = have_shadows ? ni : nullptr;
        a.nsh_out = no;
        a.region0 = reg0[h];
        a.n_regions = regn[h];
    };
    auto trace = [&](uint32_t depth, bool first, bool have_shadows, uint32_t h) {
        fill(depth, have_shadows, h);
        const uint32_t nr = regn[h];
        const uint32_t mult = depth >= 2 ? p.grid_deep : p.grid_mult;
        wf_launch_trace(s, a, p, wf_trace_grid(nr, mult, cu_trace), first, st_t);
        ++*launches;
    };
    auto shade = [&](uint32_t depth, bool first, bool have_shadows, uint32_t h) {
        fill(depth, have_shadows, h);
        hipStream_t st = st_s;
        const dim3 g(regn[h]), t(WF_SHADE_THREADS);
        if (first) {
            if (tabs)
                hipLaunchKernelGGL((k_shade<true, true>), g, t, 0, st, a);
            else
                hipLaunchKernelGGL((k_shade<true, false>), g, t, 0, st, a);
        } else {
            if (tabs)
                hipLaunchKernelGGL((k_shade<false, true>), g, t, 0, st, a);
            else
                hipLaunchKernelGGL((k_shade<false, false>), g, t, 0, st, a);
        }
        ++*launches;
    };
    auto flip = [&]() {
        std::swap(in, out);
        std::swap(shi, sho);
        std::swap(sin, sout);
        std::swap(ni, no);
    };
    if (split) {  // both masked streams start behind whatever the context's stream holds
        hipEvent_t e0 = c->sync_ev[2u * parts];
        HIPCHK(c, hipEventRecord(e0, c->stream));
        HIPCHK(c, hipStreamWaitEvent(st_t, e0, 0));
        HIPCHK(c, hipStreamWaitEvent(st_s, e0, 0));
    }
    bool flush = false;
    for (uint32_t depth = 0; depth < a.max_depth; ++depth) {
        const bool first = camera && depth == 0;
        for (uint32_t h = 0; h < parts; ++h) {
            if (split && depth > 0) HIPCHK(c, hipStreamWaitEvent(st_t, c->sync_ev[2u * h + 1u], 0));  // shade(h, depth - 1)
            trace(depth, first, depth > 0, h);
            if (split) HIPCHK(c, hipEventRecord(c->sync_ev[2u * h], st_t));
        }
        for (uint32_t h = 0; h < parts; ++h) {
            if (split) HIPCHK(c, hipStreamWaitEvent(st_s, c->sync_ev[2u * h], 0));  // trace(h, depth)
            shade(depth, first, depth > 0, h);
            if (split) HIPCHK(c, hipEventRecord(c->sync_ev[2u * h + 1u], st_s));
        }
        flip();
        HIPCHK(c, hipGetLastError());
        // unbounded depth (Mitsuba max_depth = -1): poll the live count every 8 bounces (never split: one stream)
        if (a.max_depth > 32 && (depth & 7u) == 7u) {
            std::vector<uint32_t> cnt(nreg);
            HIPCHK(c, hipMemcpyAsync(cnt.data(), sin, (size_t)nreg * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            uint64_t live = 0;
            for (uint32_t v : cnt) live += v;
            if (live == 0) {
                flush = depth + 1 < a.max_depth;  // the last bounce may have left shadow rays behind
                a.depth = depth + 1;
                break;
            }
        }
    }
    if (flush) {
        const uint32_t d = a.depth;
        trace(d, false, true, 0);
        shade(d, false, true, 0);
        flip();
        HIPCHK(c, hipGetLastError());
    }
    if (split)  // the context's stream goes on (film gather) when every part has been shaded
        for (uint32_t h = 0; h < parts; ++h) HIPCHK(c, hipStreamWaitEvent(c->stream, c->sync_ev[2u * h + 1u], 0));
    return PBRT_OK;
}
// Default paths in flight per pass of the trace / shade streams: the largest power of two (min_pass .. 512 Mi) whose workspace
// (WF_BYTES_PER_PATH each) fits two thirds of the free device memory -- what this context already holds for these buffers is
// re-used, not allocated on top --, one half of the device's total memory, and the context's workspace limit.
static uint64_t wf_default_pass_paths(pbrt_ctx *c, uint64_t min_pass) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    size_t held = 0;
    for (const char *nm : {"wf_stateA", "wf_stateB", "wf_shadowA", "wf_shadowB", "wf_hit_id", "Lhome"}) {
        auto it = c->ws.find(nm);
        if (it != c->ws.end()) held += it->second.bytes;
    }
    // two thirds of what is free, and never more than half of the device: a tenant that arrives second (torch beside the renderer,
    // USMain.py:5) still finds room, and the pass size depends less on who allocated first.  A caller that owns the device asks
    // for more per call (pbrt_film_desc.pass_paths) -- bench.py does for BASELINE config 4.
    double budget = std::min((2.0 / 3.0) * (double)(free_b + held), 0.5 * (double)total_b);
    if (c->ws_limit) budget = std::min(budget, (double)c->ws_limit - (double)(c->ws_total() - held) - 64e6 /* the small buffers */);
    uint64_t pass_paths = min_pass;
    while (pass_paths < (512u << 20) && 2.0 * (double)pass_paths * WF_BYTES_PER_PATH <= budget) pass_paths *= 2;
    return pass_paths;
}

// The guard words of the context come back with the statistics of a call (wf_guard_fetch queues the copy on the call's stream,
// wf_check_guard looks at them once the stream has drained): did a wave of k_trace run into its turn guard?
static int wf_guard_fetch(pbrt_ctx *c, uint32_t *host) {
    uint32_t *g = wf_guard(c);
    if (!g) return PBRT_E_NOMEM;
    HIPCHK(c, hipMemcpyAsync(host, g, WF_GUARD_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
    return PBRT_OK;
}
static int wf_check_guard(pbrt_ctx *c, const uint32_t *g) {
#ifdef PBRT_WF_PROBE
    {
        unsigned long long pr[8];
        std::memcpy(pr, g + 32, sizeof pr);
        fprintf(stderr, "WF_PROBE walk trips %llu lanes %llu (%.3f) holding a leaf %.3f idle %.3f | leaf trips %llu lanes %llu (%.3f) | main trips %llu rays %llu busy lanes per trip %.1f\n",
                pr[0], pr[1], pr[0] ? pr[1] / (64.0 * pr[0]) : 0.0, pr[0] ? pr[7] / (64.0 * pr[0]) : 0.0,
                pr[0] ? 1.0 - (pr[1] + pr[7]) / (64.0 * pr[0]) : 0.0, pr[2], pr[3], pr[2] ? pr[3] / (64.0 * pr[2]) : 0.0, pr[4], pr[5],
                pr[4] ? (double)pr[6] / pr[4] : 0.0);
        (void)hipMemsetAsync(wf_guard(c) + 32, 0, 64, c->stream);
    }
#endif
    if (g[0] == 0 && g[WF_GUARD_REHIT] == 0) return PBRT_OK;
    HIPCHK(c, hipMemsetAsync(wf_guard(c), 0, WF_GUARD_WORDS * 4, c->stream));
    if (g[0] == 0)
        return c->fail(PBRT_E_DEVICE, "k_shade / k_us_shade: %u hit(s) reported by k_trace were not reproduced by the repeated primitive "
                                      "test (the two must share their arithmetic and their build flags); the result is not valid",
                       g[WF_GUARD_REHIT]);
    float f[7];
    std::memcpy(f, g + 21, sizeof f);
    return c->fail(PBRT_E_DEVICE,
                   "k_trace: %u wave(s) hit the turn guard (block %u wave %u: busy %u walking %u queue_empty %u total %u q_in %u "
                   "depth 0x%x K %u | lane: cur %x sp %u tos %x rslot %x rows %x %x ovf %x %x o %g %g %g d %g %g %g best %g n_rows %u "
                   "shift %u nodes %u)",
                   g[0], g[1], g[2], g[3], g[4], g[6], g[7], g[8], g[10], g[11], g[12], g[13], g[14], g[15], g[16], g[17],
                   g[19], g[20], f[0], f[1], f[2], f[3], f[4], f[5], f[6], g[28], g[29], g[30]);
}

// Byte model of the two-launch bounce, the bytes the algorithm NEEDS, per depth d with live[d] rays of which
// hits[d] hit something (both counted on the device), S = shadow rays of the render:
//   k_trace (k_trace_primary at depth 0: the camera rays are generated in registers)
//       32 B per continuation ray (origin, direction planes; depth >= 1), 4 B hit index written (the primitive, or none)
//       per shadow ray: 32 B read + 4 B visibility written
//   k_shade
//       4 B hit index per ray ((t, u, v) are recomputed from the primitive's record, not read)
//       depth >= 1: the state of a path once -- 48 B (L / A / B planes) if its ray left the scene, all six planes (96 B) if it hit
//       16 B radiance record per path that ends, 96 B per survivor, 32 B per shadow ray it emits
// (the 64-byte primitive record and the vertex normals of a hit come from tables that stay in L2: 0 B.)  What the kernels move on
// top of that -- the 48 bytes k_shade reads twice for a path that hit, whole 128-byte lines for sparse gathers -- is traffic, not
// model: profiles/pmc_traffic.json has the ratio.  *trace = the part of the total that is k_trace's.
static uint64_t wavefront_model_bytes(const unsigned long long *live, const unsigned long long *hits, uint32_t nd, uint64_t shadows,
                                      uint64_t *trace) {
    uint64_t tr = 0, sh = 0;
    for (uint32_t d = 0; d < nd; ++d) {
        const uint64_t in = live[d], next = d + 1 < nd ? live[d + 1] : 0, h = std::min<uint64_t>(hits[d], in);
        if (!in) break;
        tr += (d > 0 ? in * 32 : 0) + in * 4;
        sh += in * 4 + (d > 0 ? (in - h) * 48 + h * 96 : 0);
        sh += (in - next) * 16 + next * 96;
    }
    tr += shadows * 36;
    sh += shadows * 32;
    if (trace) *trace = tr;
    return tr + sh;
}

static int render_impl(pbrt_scene *s, const pbrt_camera *cam, const pbrt_film_desc *f, void *d_out) {
    pbrt_ctx *c = s->ctx;
    if (int rcs = ctx_settle(c)) return rcs;
    NEED(c, cam && f && d_out);
    const uint32_t W = cam->film_w, H = cam->film_h;
    NEED(c, W > 0 && H > 0 && f->crop_w > 0 && f->crop_h > 0);
    NEED(c, (uint64_t)f->crop_x + f->crop_w <= W && (uint64_t)f->crop_y + f->crop_h <= H);
    NEED(c, f->spp > 0 && f->max_depth > 0 && f->filter <= PBRT_FILTER_GAUSSIAN);
    NEED(c, (uint64_t)W * H <= 0xffffffffull);
    HIPCHK(c, hipSetDevice(c->device));
#ifndef PBRT_DIAG
    {   // launch structures that lost their A/B live in the diagnostic build only (make -C csrc diag -> libpbrt_hip_diag.so)
        const bool bvh = s->accel_kernel == ACCEL_K_BVH_GLOBAL || s->accel_kernel == ACCEL_K_BVH_LDS;
        if ((f->flags & (PBRT_FILM_REGEN | PBRT_FILM_WALK_SET)) || (bvh && (f->flags & PBRT_FILM_NO_HIT_POOL)))
            return c->fail(PBRT_E_UNSUPPORTED, "PBRT_FILM_REGEN / PBRT_FILM_WALK_FROM / PBRT_FILM_NO_HIT_POOL select diagnostic launch "
                                               "structures: build libpbrt_hip_diag.so (make -C csrc diag)");
    }
#endif
    int rc = set_lds_attr(s);
    if (rc) return rc;
    const uint32_t R = f->filter == PBRT_FILTER_BOX ? 0 : (f->filter == PBRT_FILTER_TENT ? 1 : 2);
    const uint32_t rx0 = f->crop_x > R ? f->crop_x - R : 0, ry0 = f->crop_y > R ? f->crop_y - R : 0;
    const uint32_t rx1 = std::min(f->crop_x + f->crop_w + R, W), ry1 = std::min(f->crop_y + f->crop_h + R, H);
    const uint32_t rw = rx1 - rx0, rh = ry1 - ry0;
    const uint64_t npix_r = (uint64_t)rw * rh;
    const uint64_t film_px = (uint64_t)f->crop_w * f->crop_h;
    // default paths in flight per pass, measured on cbox 512^2 x 256 (one box): 2 / 4 / 8 / 16 / 32 / 64 Mi -> 9.55 / 8.56 / 8.14 / 7.92 /
    // 8.13 / 8.20 ms (fewer launch tails against cache residency of the ping-pong state)
    // round 2, fused first launch: 2 / 4 / 8 / 16 / 32 / 64 Mi -> 9.09 / 8.07 / 7.65 / 7.37 / 7.31 / 7.39 ms; BVH scenes keep 16 Mi
    const bool brute_scene = s->accel_kernel == ACCEL_K_BRUTE || s->accel_kernel == ACCEL_K_BRUTE_BIG;
    // round 2, chains of up to six bounces per launch (one launch per pass on the Cornell box): 2 / 4 / 8 / 16 / 32 / 64 Mi -> 7.85 / 7.01 /
    // 6.72 / 6.56 / 6.49 / 6.43 ms
    // round 3, BVH scenes as trace / shade streams (twelve launches per pass: their tails and the thinly filled late bounces weigh
    // less in larger passes): ring 1024^2 x 64: 1 / 2 / 4 / 8 / 16 / 32 / 64 Mi -> 50.8 / 37.6 / 25.5 / 22.2 / 19.6 / 18.0 / 17.3 ms;
    // 64 Mi paths = 23 GB of workspace (356 B per path in flight then, 340 B since round 4); the fused BVH kernels (PBRT_FILM_NO_HIT_POOL) keep 16 Mi
    const bool wf_scene = !brute_scene && !(f->flags & PBRT_FILM_NO_HIT_POOL);
    // and beyond: 1024^2 x 512: 64 / 128 / 256 Mi -> 144 / 134 / 115 ms.  Default for BVH scenes: the largest power of two whose
    // workspace (WF_BYTES_PER_PATH = 340 B per path in flight, allocated as asked) fits two thirds of the free device memory and
    // the context's workspace limit, 1 .. 512 Mi (round 4, 1024^2 x 512: 128 / 256 / 512 Mi -> 105.9 / 99.5 / 94.6 ms; 512 Mi paths =
    // 183 GB of the 288 GB of an MI355X -- a renderer that owns the device takes it; one that shares it sets a limit, see
    // pbrt_ctx_set_workspace_limit, and pbrt_ctx_trim hands the memory back).  The free-memory figure is a snapshot
    // (another process may allocate between the query and the hipMalloc), so a failed allocation halves the pass and tries again.
    // (pbrt_ctx::call_seq was bumped by the entry point, before its first workspace request.)
    const uint64_t WF_MIN_PASS = 1u << 20;
    uint64_t pass_paths = f->pass_paths ? f->pass_paths : (brute_scene ? (64u << 20) : (16u << 20));
    if (!f->pass_paths && wf_scene) {
        pass_paths = wf_default_pass_paths(c, WF_MIN_PASS);
    } else if (!f->pass_paths && c->ws_limit) {
        // brute-force scenes under a workspace limit: 16 B of radiance record per path, and the ping-pong state (2 x 60 B) if the
        // launch plan turns out to need it
        const double budget = (double)c->ws_limit - 64e6;
        while (pass_paths > WF_MIN_PASS && (double)pass_paths * (16 + 2 * N_STATE * 4) > budget) pass_paths /= 2;
    }
    // BVH scenes: intersection and shading as separate streams (kernels_wavefront.h); PBRT_FILM_NO_HIT_POOL keeps the fused
    // k_bounce (one launch per bounce, shading in the lanes the traversal leaves) as the A/B reference
    const bool bvh_scene = s->accel_kernel == ACCEL_K_BVH_GLOBAL || s->accel_kernel == ACCEL_K_BVH_LDS;
    const bool wavefront = bvh_scene && !(f->flags & PBRT_FILM_NO_HIT_POOL);
    static_assert(WF_REGION == REGION_SEGS_BVH * SEG_BVH, "both BVH launch structures cut a pass into the same regions");
    const uint32_t REGION = rad_region_segs(s->accel_kernel) * seg_threads(s->accel_kernel);
    uint32_t s_pass = 1, cap = 0, nseg = 0;
    WfBufs wfb{};
    WfPlan wfp;
    float *Lhome = nullptr;
    for (;;) {
        s_pass = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(f->spp, pass_paths / std::max<uint64_t>(npix_r, 1)));
        s_pass = div_up(f->spp, div_up(f->spp, s_pass));  // equal passes instead of full ones plus a small remainder
        NEED(c, npix_r * s_pass < 0xfffffc00ull);
        cap = div_up(npix_r * s_pass, REGION) * REGION;
        nseg = cap / REGION;  // regions (one workgroup each)
        NEED(c, wf_scene || (uint64_t)cap * N_STATE * 4 < 0xffffffffull);  // the tiled state arrays are addressed through 32-bit buffer offsets
        if (wavefront) NEED(c, cap < WF_DEAD);  // ray records are addressed with two flag bits on top
        Lhome = (float *)c->buf("Lhome", (size_t)cap * 16);  // float4 (r, g, b, 0) per home
        if (Lhome && (!wavefront || wf_alloc(c, cap, nseg, &wfb))) break;
        // out of memory (or over the context's limit): give the pass buffers back and try with half the paths in flight
        if (f->pass_paths || s_pass <= 1 || pass_paths <= WF_MIN_PASS) {
            release_pass_buffers(c);  // what did fit goes back too: the caller may be about to give the memory to someone else
            return PBRT_E_NOMEM;
        }
        release_pass_buffers(c);
        pass_paths = std::max<uint64_t>(WF_MIN_PASS, std::min<uint64_t>(pass_paths, npix_r * s_pass) / 2);
    }
    if (wavefront) {
        wfp = wf_plan(s);
        if ((rc = wf_set_attr(s, wfp)) != 0) return rc;
    }
    // Brute-force scenes: the ping-pong path state (2 x 60 B per slot, 8 GB at 64 Mi paths) is only touched by a pass that needs
    // more than one bounce launch; with the plan that walks every bounce in one launch (the Cornell box) it is never written, so it
    // is allocated per pass, for the paths of THAT pass, when its plan says so (state_for below).
    float *stA = wavefront ? (float *)wfb.stA : nullptr;
    float *stB = wavefront ? (float *)wfb.stB : nullptr;
    // fused BVH kernels: the live paths are made dense again before every bounce of depth >= 2 (k_scan_owners / k_repack_copy)
    const bool repack = bvh_scene && !wavefront && rad_wave_private(s->accel_kernel) && !(f->flags & PBRT_FILM_NO_REPACK);
    float *stC = repack ? (float *)c->buf("stateC", (size_t)cap * N_STATE * 4) : nullptr;
    // live counters and statistics rows: one per region, or one per wave of it (BVH kernels: wave-private compaction)
    const uint32_t n_own = nseg * rad_owners_per_region(s->accel_kernel);
    uint32_t *segA = (uint32_t *)c->buf("segA", (size_t)n_own * 4);
    uint32_t *segB = (uint32_t *)c->buf("segB", (size_t)n_own * 4);
    uint32_t *segC = repack ? (uint32_t *)c->buf("segC", (size_t)n_own * 4) : nullptr;
    uint32_t *offs = repack ? (uint32_t *)c->buf("seg_offs", (size_t)n_own * 4) : nullptr;
    uint32_t *quota = repack ? (uint32_t *)c->buf("seg_quota", 64) : nullptr;
    float *acc = (float *)c->buf("film_acc", film_px * 16);
    unsigned long long *dstats = (unsigned long long *)c->buf("stats", (2 + 2 * MAX_DEPTH_STATS) * 8);
    const uint32_t n_rows = nseg * (wavefront ? WF_SHADE_THREADS / 64u : rad_rows_per_region(s->accel_kernel));  // statistics rows
    // statistics rows in use: segments, shadow rays, one per depth (cleared and reduced per call: keep it to what the call touches)
    const uint32_t stat_rows = 2 + (uint32_t)std::min<uint64_t>(f->max_depth, MAX_DEPTH_STATS);
    const size_t segstats_bytes = (size_t)stat_rows * n_rows * 8;  // reduced at the end
    unsigned long long *segstats = (unsigned long long *)c->buf("segstats", (size_t)(2 + 2 * MAX_DEPTH_STATS) * n_rows * 8);
    if (!segstats) return PBRT_E_NOMEM;
    if (!Lhome || !segA || !segB || !acc || !dstats) return PBRT_E_NOMEM;
    // k_bounce_pool launches also count the rays of every depth that hit something (rows HIT_ROW0 + d, for the byte model)
    const bool hit_pool = wavefront;  // k_shade counts the rays of every depth that hit something (rows HIT_ROW0 + d, for the byte model)
    const uint32_t hit_rows = hit_pool ? stat_rows - 2 : 0;
    if (repack && (!stC || !segC || !offs || !quota)) return PBRT_E_NOMEM;
    hipStream_t st = c->stream;
    HIPCHK(c, hipMemsetAsync(acc, 0, film_px * 16, st));
    HIPCHK(c, hipMemsetAsync(dstats, 0, (2 + 2 * MAX_DEPTH_STATS) * 8, st));
    HIPCHK(c, hipMemsetAsync(segstats, 0, segstats_bytes, st));
    if (hit_rows) HIPCHK(c, hipMemsetAsync(segstats + (size_t)HIT_ROW0 * n_rows, 0, (size_t)hit_rows * n_rows * 8, st));
    HIPCHK(c, hipEventRecord(c->ev0, st));
    size_t n_ev = 0;
    hipEvent_t pass_e1 = nullptr;
    uint32_t passes = 0, launches = 0;
    // the fuse plan of this call: the caller's, or the one learnt from the last render of this scene, or the library default
    const bool brute_scene_k = s->accel_kernel == ACCEL_K_BRUTE || s->accel_kernel == ACCEL_K_BRUTE_BIG;
    const bool plan_was_learnt = brute_scene_k && s->plan_hint_valid;
    uint32_t call_plan = !brute_scene_k ? 0u
                         : (f->flags & PBRT_FILM_FUSE_PLAN_SET) ? ((f->flags >> 8) & 0xffu)
                         : (s->plan_hint_valid ? s->plan_hint : PBRT_DEFAULT_FUSE_PLAN);
    // First render of a brute-force scene with the plan left to the library: the first PROBE_SPP samples are a pass of their own,
    // their path survival is read back (one synchronisation, ~30 us) and decides the plan of all the other passes.  The film does
    // not depend on how the samples are split into passes, nor on the plan.
    constexpr uint32_t PROBE_SPP = 2;
    const bool probe = brute_scene_k && !s->plan_hint_valid && f->spp >= 8 * PROBE_SPP && s_pass >= PROBE_SPP &&
                       !(f->flags & (PBRT_FILM_FUSE_PLAN_SET | PBRT_FILM_WALK_SET | PBRT_FILM_REGEN));
    uint32_t s_step = s_pass;  // samples of a regular pass
    for (uint32_t s0 = 0; s0 < f->spp; ++passes) {
        const uint32_t sc = (probe && passes == 0) ? PROBE_SPP : std::min(s_step, f->spp - s0);
        RadArgs a{};
        a.sc = s->ds;
        if ((f->flags & PBRT_FILM_NO_OCCLUDER_PRUNING) && a.sc.occ_prims) {  // diagnostic: shadow segments walk every primitive
            a.sc.occ_prims = a.sc.prims;
            a.sc.n_occ = a.sc.n_prims;
        }
        a.cam = *cam;
        a.Lhome = Lhome;
        a.stats = segstats;
        a.stat_stride = n_rows;
        a.cap = cap;
        a.state_cap = cap;
        a.n_paths = (uint32_t)(npix_r * sc);
        a.max_depth = f->max_depth;
        a.rr_depth = f->rr_depth;
        a.seed = f->seed;
        a.key_mode = 0;
        a.rx0 = rx0;
        a.ry0 = ry0;
        a.rw = rw;
        a.npix_r = (uint32_t)npix_r;
        a.tile_rows = rh & ~7u;
        a.div_npix = make_fastdiv(a.npix_r);
        a.div_rw = make_fastdiv(rw);
        for (uint32_t n : {0u, 1u, rw - 1, rw, rw + 1, a.npix_r - 1, a.npix_r, a.npix_r + 1, cap - 1, cap, 0xffffffffu}) {
            NEED(c, udiv_fast_host(n, a.div_npix) == n / a.npix_r && udiv_fast_host(n, a.div_rw) == n / rw);
        }
        a.s_first = f->sample_offset + s0;
        a.film_w = W;
        a.film_h = H;
        a.lds_bytes = s->lds_bytes;
        const uint32_t nseg_pass = div_up(a.n_paths, REGION);
        if (!wavefront) {  // state for this pass: none if one launch walks all its bounces
            const bool one_launch = brute_scene_k && !(f->flags & (PBRT_FILM_WALK_SET | PBRT_FILM_REGEN)) &&
                                    chain_len(call_plan, 0, f->max_depth) >= f->max_depth;
            const size_t need = one_launch ? (size_t)REGION : (size_t)nseg_pass * REGION;
            stA = (float *)c->buf("stateA", need * N_STATE * 4);
            stB = (float *)c->buf("stateB", need * N_STATE * 4);
            if (!stA || !stB) return PBRT_E_NOMEM;
            a.state_cap = (uint32_t)need;
        }
        float *in = stA, *out = stB;
        uint32_t *sin = segA, *sout = segB;
        const bool brute = s->accel_kernel == ACCEL_K_BRUTE || s->accel_kernel == ACCEL_K_BRUTE_BIG;
        const uint32_t fuse_plan = call_plan;
        const uint32_t walk_from = (f->flags & PBRT_FILM_WALK_SET) ? ((f->flags >> 17) & 0xffu) : PBRT_DEFAULT_WALK_FROM;
        if (wavefront) {
            WfArgs w{};
            w.sc = a.sc;
            w.cam = a.cam;
            w.Lhome = Lhome;
            w.stats = segstats;
            w.stat_stride = n_rows;
            w.cap = cap;
            w.n_paths = a.n_paths;
            w.max_depth = a.max_depth;
            w.rr_depth = a.rr_depth;
            w.seed = a.seed;
            w.key_mode = 0;
            w.rx0 = a.rx0;
            w.ry0 = a.ry0;
            w.rw = a.rw;
            w.npix_r = a.npix_r;
            w.s_first = a.s_first;
            w.film_w = a.film_w;
            w.film_h = a.film_h;
            w.tile_rows = a.tile_rows;
            w.div_npix = a.div_npix;
            w.div_rw = a.div_rw;
            pass_e1 = c->event(n_ev + 1);
            hipEvent_t e0 = c->event(n_ev);
            if (!e0 || !pass_e1) return c->fail(PBRT_E_DEVICE, "hipEventCreate failed");
            n_ev += 2;
            HIPCHK(c, hipEventRecord(e0, st));
            if ((rc = wf_bounces(s, w, wfb, wfp, nseg_pass, true, &launches)) != 0) return rc;
        }
#ifdef PBRT_DIAG
        else if (brute && (f->flags & PBRT_FILM_REGEN)) {
            // persistent waves with path regeneration (k_regen): one launch per pass, as many workgroups as the GPU holds at once
            int per_cu = 0;
            const void *fn = s->accel_kernel == ACCEL_K_BRUTE ? reinterpret_cast<const void *>(&k_regen<ACCEL_K_BRUTE>)
                                                              : reinterpret_cast<const void *>(&k_regen<ACCEL_K_BRUTE_BIG>);
            HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, REGEN_WG, 0));
            const uint32_t wpw = REGEN_WG / 64;
            uint32_t grid = (uint32_t)std::max(per_cu, 1) * (uint32_t)c->n_cu;
            grid = std::min(grid, std::max(1u, n_rows / wpw));            // one statistics row per wave
            grid = std::min(grid, div_up(a.n_paths, REGEN_WG));
            a.depth = 0;
            pass_e1 = c->event(n_ev + 1);
            hipEvent_t e0 = c->event(n_ev);
            if (!e0 || !pass_e1) return c->fail(PBRT_E_DEVICE, "hipEventCreate failed");
            n_ev += 2;
            HIPCHK(c, hipEventRecord(e0, st));
            if (s->accel_kernel == ACCEL_K_BRUTE)
                hipLaunchKernelGGL(k_regen<ACCEL_K_BRUTE>, dim3(grid), dim3(REGEN_WG), 0, st, a);
            else
                hipLaunchKernelGGL(k_regen<ACCEL_K_BRUTE_BIG>, dim3(grid), dim3(REGEN_WG), 0, st, a);
            HIPCHK(c, hipGetLastError());
            ++launches;
        }
#endif
        else
        for (uint32_t depth = 0; depth < f->max_depth;) {
            // bounces this launch walks: 2 at the depths of the fuse plan (brute-force kernels; the last bounce of a
            // path only looks for emitters, so it is never worth a launch slot of its own either)
            const uint32_t nb = brute ? chain_len(fuse_plan, depth, f->max_depth) : 1u;
            a.depth = depth;
            a.nb = nb;
            a.repack_mask = chain_repack_mask();
            a.in = in;
            a.out = out;
            a.seg_in = sin;
            a.seg_out = sout;
#ifdef PBRT_DIAG
            if (repack && depth >= 2) {
                // Only the regions THIS pass launched take part: a short last pass (spp not a multiple of the pass size)
                // launches nseg_pass < nseg workgroups, so paths dealt to regions >= nseg_pass would never be traced, and
                // the counters of those regions are left over from the previous pass.
                const uint32_t owners = rad_owners_per_region(s->accel_kernel), wreg = REGION / owners;
                const uint32_t n_own_pass = nseg_pass * owners;
                hipLaunchKernelGGL(k_scan_owners, dim3(1), dim3(1024), 0, st, sin, n_own_pass, wreg, owners, (uint32_t)c->n_cu, offs,
                                   segC, quota);
                hipLaunchKernelGGL(k_repack_copy, dim3(n_own_pass), dim3(256), 0, st, in, stC, sin, offs, quota, n_own_pass, wreg);
                a.in = stC;
                a.seg_in = segC;
            }
#endif
            // ONE event pair per pass around its bounce launches (a pair per launch costs ~8 us of queue bubbles each)
            if (depth == 0) {
                pass_e1 = c->event(n_ev + 1);
                hipEvent_t e0 = c->event(n_ev);
                if (!e0 || !pass_e1) return c->fail(PBRT_E_DEVICE, "hipEventCreate failed");
                n_ev += 2;
                HIPCHK(c, hipEventRecord(e0, st));
            }
            const bool walk = brute && depth >= walk_from;  // this launch walks every remaining bounce of the pass
#ifdef PBRT_DIAG
            const uint32_t pair_m = (brute && depth == 0 && !walk) ? pair_merge_bounce(fuse_plan, f->max_depth) : 0u;
            if (pair_m) {  // the whole path in one launch, two tiles per wave
                a.merge_at = pair_m;
                const uint32_t g2 = div_up(a.n_paths, 2u * SEG_BRUTE);
                if (s->accel_kernel == ACCEL_K_BRUTE)
                    hipLaunchKernelGGL(k_chain_pair<ACCEL_K_BRUTE>, dim3(g2), dim3(SEG_BRUTE), 0, st, a);
                else
                    hipLaunchKernelGGL(k_chain_pair<ACCEL_K_BRUTE_BIG>, dim3(g2), dim3(SEG_BRUTE), 0, st, a);
            } else if (walk) {
                if (depth == 0)
                    launch_walk<true>(s, a, nseg_pass, nb);
                else
                    launch_walk<false>(s, a, nseg_pass, nb);
            } else
#endif
            if ((rc = depth == 0 ? launch_bounce<true>(s, a, nseg_pass, nb) : launch_bounce<false>(s, a, nseg_pass, nb)) != 0) return rc;
            HIPCHK(c, hipGetLastError());
            ++launches;
            if (walk) break;
            std::swap(in, out);
            std::swap(sin, sout);
            const uint32_t depth_before = depth;
            depth += nb;
            // unbounded depth (Mitsuba max_depth = -1): poll the live count every 8 bounces
            if (f->max_depth > 32 && (depth >> 3) != (depth_before >> 3)) {
                std::vector<uint32_t> cnt(n_own);
                HIPCHK(c, hipMemcpyAsync(cnt.data(), sin, (size_t)n_own * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
                uint64_t live = 0;
                for (uint32_t v : cnt) live += v;
                if (live == 0) break;
            }
        }
        HIPCHK(c, hipEventRecord(pass_e1, st));
        FilmArgs fa{};
        fa.Lhome = Lhome;
        fa.acc = acc;
        fa.cap = cap;
        fa.cx0 = f->crop_x;
        fa.cy0 = f->crop_y;
        fa.cw = f->crop_w;
        fa.ch = f->crop_h;
        fa.rx0 = rx0;
        fa.ry0 = ry0;
        fa.rw = rw;
        fa.rh = rh;
        fa.npix_r = (uint32_t)npix_r;
        fa.tile_rows = a.tile_rows;
        fa.s_first = a.s_first;
        fa.s_count = sc;
        fa.film_w = W;
        fa.film_h = H;
        fa.filter = f->filter;
        fa.seed = f->seed;
        if (f->filter == PBRT_FILTER_BOX)
            hipLaunchKernelGGL(k_film_accum, dim3(div_up(film_px, 256)), dim3(256), 0, st, fa);
        else {
            // small crops: 4 x as many (single-wave) workgroups; same sums, pixel by pixel
            const bool big = (uint64_t)div_up(f->crop_w, 16) * div_up(f->crop_h, 16) >= 4ull * (uint64_t)c->n_cu;
            const bool tent = f->filter == PBRT_FILTER_TENT;
            const dim3 g16(div_up(f->crop_w, 16), div_up(f->crop_h, 16)), g8(div_up(f->crop_w, 8), div_up(f->crop_h, 8));
            if (big && tent)
                hipLaunchKernelGGL((k_film_accum_tiled<16, PBRT_FILTER_TENT>), g16, dim3(16, 16), 0, st, fa);
            else if (big)
                hipLaunchKernelGGL((k_film_accum_tiled<16, PBRT_FILTER_GAUSSIAN>), g16, dim3(16, 16), 0, st, fa);
            else if (tent)
                hipLaunchKernelGGL((k_film_accum_tiled<8, PBRT_FILTER_TENT>), g8, dim3(8, 8), 0, st, fa);
            else
                hipLaunchKernelGGL((k_film_accum_tiled<8, PBRT_FILTER_GAUSSIAN>), g8, dim3(8, 8), 0, st, fa);
        }
        HIPCHK(c, hipGetLastError());
        s0 += sc;
        if (probe && passes == 0) {  // learn the plan from the probe pass
            unsigned long long hp[2 + MAX_DEPTH_STATS];
            hipLaunchKernelGGL(k_reduce_stats, dim3(stat_rows, REDUCE_SLICES), dim3(256), 0, st, segstats, n_rows, (size_t)n_rows, dstats);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(hp, dstats, (size_t)stat_rows * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            HIPCHK(c, hipMemsetAsync(dstats, 0, (2 + 2 * MAX_DEPTH_STATS) * 8, st));  // the final reduction adds every row again
            for (uint32_t d = stat_rows; d < 2 + MAX_DEPTH_STATS; ++d) hp[d] = 0;
            call_plan = plan_from_survival(hp + 2, (uint32_t)std::min<uint64_t>(f->max_depth, MAX_DEPTH_STATS));
            const uint32_t rem = f->spp - s0;
            s_step = div_up(rem, div_up(rem, s_pass));  // equal passes for what is left
        }
    }
    hipLaunchKernelGGL(k_film_resolve, dim3(div_up(film_px, 256)), dim3(256), 0, st, acc, (float *)d_out, (uint32_t)film_px,
                       (uint32_t)((f->flags & PBRT_FILM_RAW_ACCUM) ? 1 : 0));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev1, st));
    unsigned long long hstats[2 + 2 * MAX_DEPTH_STATS];
    hipLaunchKernelGGL(k_reduce_stats, dim3(stat_rows, REDUCE_SLICES), dim3(256), 0, st, segstats, n_rows, (size_t)n_rows, dstats);
    if (hit_rows)
        hipLaunchKernelGGL(k_reduce_stats, dim3(hit_rows, REDUCE_SLICES), dim3(256), 0, st, segstats + (size_t)HIT_ROW0 * n_rows, n_rows,
                           (size_t)n_rows, dstats + HIT_ROW0);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hstats, dstats, sizeof hstats, hipMemcpyDeviceToHost, st));
    uint32_t hguard[WF_GUARD_WORDS] = {0};
    if (wavefront && (rc = wf_guard_fetch(c, hguard)) != 0) return rc;
    HIPCHK(c, hipStreamSynchronize(st));
    if (wavefront && (rc = wf_check_guard(c, hguard)) != 0) return rc;
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    double bounce_ms = 0.0;
    for (size_t i = 0; i + 1 < n_ev; i += 2) {
        float t = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&t, c->ev_pool[i], c->ev_pool[i + 1]));
        bounce_ms += t;
    }
    pbrt_stats &S = c->stats;
    S = pbrt_stats{};
    S.samples = npix_r * f->spp;
    S.segments = hstats[0];
    S.shadow_rays = hstats[1];
    S.kernel_ms = ms;
    S.bounce_ms = bounce_ms;
    S.bounce_launches = launches;
    S.passes = passes;
    S.fuse_plan = call_plan;
    S.plan_source = wavefront ? PBRT_PLAN_STREAMS
                    : (f->flags & PBRT_FILM_FUSE_PLAN_SET) ? PBRT_PLAN_CALLER
                    : probe ? PBRT_PLAN_PROBED
                    : plan_was_learnt ? PBRT_PLAN_LEARNT : PBRT_PLAN_DEFAULT;
    S.pass_paths = (uint64_t)npix_r * s_pass;
    S.workspace_bytes = 0;
    for (const auto &kv : c->ws) S.workspace_bytes += kv.second.bytes;
    uint64_t tot, bb;
    for (int d = 0; d < 16; ++d) S.live[d] = hstats[2 + d];
    const bool brute_k = s->accel_kernel == ACCEL_K_BRUTE || s->accel_kernel == ACCEL_K_BRUTE_BIG;
    const uint32_t plan = call_plan;
    if (brute_k && hstats[2] > 0) {  // remember what this scene's paths do for the next render of it
        s->plan_hint = plan_from_survival(hstats + 2, (uint32_t)std::min<uint64_t>(f->max_depth, MAX_DEPTH_STATS));
        s->plan_hint_valid = true;
    }
    radiance_model_bytes(hstats + 2, MAX_DEPTH_STATS, S.samples, film_px, passes, plan, f->max_depth,
                         hit_pool ? hstats + HIT_ROW0 : nullptr, &tot, &bb);
    if (wavefront) {
        tot -= bb;
        uint64_t trb = 0;
        bb = wavefront_model_bytes(hstats + 2, hstats + HIT_ROW0, MAX_DEPTH_STATS, S.shadow_rays, &trb);
        tot += bb;
        S.trace_model_bytes = trb;
    }
    if (brute_k && (f->flags & PBRT_FILM_REGEN)) {  // k_regen keeps the paths in registers: only the radiance records are written
        tot -= bb;
        bb = S.samples * 12;
        tot += bb;
    }
    S.model_bytes = tot;
    S.bounce_model_bytes = bb;
    return PBRT_OK;
}

#ifdef PBRT_BVH_PROBE  // diagnostic builds only (tools/bvh_probe.py); not part of the ABI
extern "C" int pbrt_debug_bvh_probe(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bvh_probe), 64) != hipSuccess) return PBRT_E_DEVICE;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_probe), z, 64) != hipSuccess) return PBRT_E_DEVICE;
    }
    return PBRT_OK;
}
#endif

extern "C" {

int pbrt_render_radiance_dev(pbrt_scene *s, const pbrt_camera *cam, const pbrt_film_desc *f, void *d_out) {
    if (!s) return PBRT_E_INVALID;
    ++s->ctx->call_seq;
    return render_impl(s, cam, f, d_out);
}

int pbrt_render_radiance(pbrt_scene *s, const pbrt_camera *cam, const pbrt_film_desc *f, float *out) {
    if (!s) return PBRT_E_INVALID;
    pbrt_ctx *c = s->ctx;
    NEED(c, cam && f && out);
    HIPCHK(c, hipSetDevice(c->device));
    ++c->call_seq;
    const size_t n = (size_t)f->crop_w * f->crop_h * ((f->flags & PBRT_FILM_RAW_ACCUM) ? 4 : 3);
    void *d = c->buf("film_out", n * 4);
    if (!d) return PBRT_E_NOMEM;
    int rc = render_impl(s, cam, f, d);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(out, d, n * 4, hipMemcpyDeviceToHost));
    return PBRT_OK;
}

int pbrt_integrator_sample(pbrt_scene *s, uint32_t n, const float *o, const float *d, const float *tmax,
                           uint32_t index_offset, uint32_t sample_index, uint32_t seed, uint32_t max_depth,
                           uint32_t rr_depth, float *rgb) {
    if (!s) return PBRT_E_INVALID;
    pbrt_ctx *c = s->ctx;
    NEED(c, o && d && tmax && rgb && max_depth > 0);
    if (n == 0) return PBRT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rcs = ctx_settle(c)) return rcs;
    int rc = set_lds_attr(s);
    if (rc) return rc;
    ++c->call_seq;
    const uint32_t REGION = rad_region_segs(s->accel_kernel) * seg_threads(s->accel_kernel);
    const uint32_t cap = div_up(n, REGION) * REGION, nseg = cap / REGION;
    if (s->accel_kernel == ACCEL_K_BVH_GLOBAL || s->accel_kernel == ACCEL_K_BVH_LDS) {  // trace / shade streams
        NEED(c, cap < WF_DEAD);
        WfBufs b{};
        if (!wf_alloc(c, cap, nseg, &b)) return PBRT_E_NOMEM;
        const WfPlan p = wf_plan(s);
        if ((rc = wf_set_attr(s, p)) != 0) return rc;
        float *Lh = (float *)c->buf("Lhome", (size_t)cap * 16);
        const uint32_t n_rows = nseg * (WF_SHADE_THREADS / 64u);
        unsigned long long *rows = (unsigned long long *)c->buf("segstats", (size_t)(2 + 2 * MAX_DEPTH_STATS) * n_rows * 8);
        float *io = (float *)c->buf("leaf_io", (size_t)n * 7 * 4);
        if (!Lh || !rows || !io) return PBRT_E_NOMEM;
        hipStream_t st = c->stream;
        HIPCHK(c, hipMemcpyAsync(io, o, (size_t)n * 12, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(io + 3 * (size_t)n, d, (size_t)n * 12, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(io + 6 * (size_t)n, tmax, (size_t)n * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemsetAsync(rows, 0, (size_t)(2 + 2 * MAX_DEPTH_STATS) * n_rows * 8, st));
        HIPCHK(c, hipMemsetAsync(Lh, 0, (size_t)cap * 16, st));
        hipLaunchKernelGGL(k_init_rays_wf, dim3(div_up(std::max(n, nseg), 256)), dim3(256), 0, st, b.stA, cap, b.segA, nseg, n, io,
                           io + 3 * (size_t)n, io + 6 * (size_t)n);
        WfArgs w{};
        w.sc = s->ds;
        w.Lhome = Lh;
        w.stats = rows;
        w.stat_stride = n_rows;
        w.cap = cap;
        w.n_paths = n;
        w.max_depth = max_depth;
        w.rr_depth = rr_depth;
        w.seed = seed;
        w.key_mode = 1;
        w.npix_r = 1;
        w.rw = 1;
        w.div_npix = w.div_rw = make_fastdiv(1);
        w.index_offset = index_offset;
        w.sample_index = sample_index;
        uint32_t launches = 0;
        if ((rc = wf_bounces(s, w, b, p, nseg, false, &launches)) != 0) return rc;
        uint32_t hguard[WF_GUARD_WORDS] = {0};
        if ((rc = wf_guard_fetch(c, hguard)) != 0) return rc;
        HIPCHK(c, hipStreamSynchronize(st));
        if ((rc = wf_check_guard(c, hguard)) != 0) return rc;
        std::vector<float> rec((size_t)n * 4);
        HIPCHK(c, hipMemcpy(rec.data(), Lh, (size_t)n * 16, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) rgb[(size_t)k * n + i] = rec[(size_t)i * 4 + k];
        return PBRT_OK;
    }
    float *stA = (float *)c->buf("stateA", (size_t)cap * N_STATE * 4);
    float *stB = (float *)c->buf("stateB", (size_t)cap * N_STATE * 4);
    float *Lhome = (float *)c->buf("Lhome", (size_t)cap * 16);  // float4 (r, g, b, 0) per home
    const uint32_t owners = rad_owners_per_region(s->accel_kernel), n_own = nseg * owners;  // see render_impl
    uint32_t *segA = (uint32_t *)c->buf("segA", (size_t)n_own * 4);
    uint32_t *segB = (uint32_t *)c->buf("segB", (size_t)n_own * 4);
    unsigned long long *dstats = (unsigned long long *)c->buf("stats", (2 + MAX_DEPTH_STATS) * 8);
    const uint32_t n_rows = nseg * rad_rows_per_region(s->accel_kernel);
    const size_t segstats_bytes = (size_t)(2 + MAX_DEPTH_STATS) * n_rows * 8;  // rows, reduced at the end
    unsigned long long *segstats = (unsigned long long *)c->buf("segstats", segstats_bytes);
    if (!segstats) return PBRT_E_NOMEM;
    float *io = (float *)c->buf("leaf_io", (size_t)n * 7 * 4);
    if (!stA || !stB || !Lhome || !segA || !segB || !dstats || !io) return PBRT_E_NOMEM;
    hipStream_t st = c->stream;
    HIPCHK(c, hipMemcpyAsync(io, o, (size_t)n * 12, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(io + 3 * (size_t)n, d, (size_t)n * 12, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(io + 6 * (size_t)n, tmax, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(dstats, 0, (2 + MAX_DEPTH_STATS) * 8, st));
    HIPCHK(c, hipMemsetAsync(segstats, 0, segstats_bytes, st));
    HIPCHK(c, hipMemsetAsync(Lhome, 0, (size_t)cap * 16, st));
    hipLaunchKernelGGL(k_init_rays, dim3(div_up(std::max(n, n_own), 256)), dim3(256), 0, st, stA, segA, n_own, REGION / owners, n, io,
                       io + 3 * (size_t)n, io + 6 * (size_t)n);
    RadArgs a{};
    a.sc = s->ds;
    a.Lhome = Lhome;
    a.stats = segstats;
    a.stat_stride = n_rows;
    a.cap = cap;
    a.state_cap = cap;
    a.n_paths = n;
    a.max_depth = max_depth;
    a.rr_depth = rr_depth;
    a.seed = seed;
    a.key_mode = 1;
    a.npix_r = 1;
    a.rw = 1;
    a.div_npix = a.div_rw = make_fastdiv(1);
    a.index_offset = index_offset;
    a.sample_index = sample_index;
    a.lds_bytes = s->lds_bytes;
    float *in = stA, *out = stB;
    uint32_t *sin = segA, *sout = segB;
    for (uint32_t depth = 0; depth < max_depth; ++depth) {
        a.depth = depth;
        a.in = in;
        a.out = out;
        a.seg_in = sin;
        a.seg_out = sout;
        if ((rc = launch_bounce<false>(s, a, nseg)) != 0) return rc;
        HIPCHK(c, hipGetLastError());
        std::swap(in, out);
        std::swap(sin, sout);
        if (max_depth > 32 && (depth & 7) == 7) {
            std::vector<uint32_t> cnt(n_own);
            HIPCHK(c, hipMemcpyAsync(cnt.data(), sin, (size_t)n_own * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            uint64_t live = 0;
            for (uint32_t v : cnt) live += v;
            if (live == 0) break;
        }
    }
    HIPCHK(c, hipStreamSynchronize(st));
    std::vector<float> rec((size_t)n * 4);
    HIPCHK(c, hipMemcpy(rec.data(), Lhome, (size_t)n * 16, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) rgb[(size_t)k * n + i] = rec[(size_t)i * 4 + k];
    return PBRT_OK;
}

// ------------------------------------------------------------------------------------------------
// ultrasound mode driver
// ------------------------------------------------------------------------------------------------
static inline float host_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

int pbrt_us_tx_delays(const pbrt_us_params *p, float *tx) {
    if (!p || !tx || p->n_angles > PBRT_US_MAX_ANGLES || p->n_angles == 0 || p->n_elements == 0) return PBRT_E_INVALID;
    for (uint32_t a = 0; a < p->n_angles; ++a) {
        double ar = (double)p->angles_deg[a] * (M_PI / 180.0);  // np.deg2rad, CustomIntegrator.py:247
        for (uint32_t e = 0; e < p->n_elements; ++e) {
            // :248 float32 elem_x = pitch * (arange_f32 - (N-1)/2); :254,257 tx = f32(elem_x * sin(a) / c)
            float ex = (float)((double)p->pitch * ((double)(float)e - ((double)p->n_elements - 1.0) / 2.0));
            tx[a * p->n_elements + e] = (float)(((double)ex * std::sin(ar)) / (double)p->sound_speed);
        }
    }
    return PBRT_OK;
}

}  // extern "C"

// which instance of k_us_bounce: the switches of the library's default (with / without the carrier) are compiled in, any other set
// -- and PBRT_US_GENERIC_KERNEL=1 -- takes the instance that reads them at run time (kernels_us.h)
static uint32_t us_kernel_quirks(const UsArgs &a) {
    const char *gen = getenv("PBRT_US_GENERIC_KERNEL");  // read per launch: a test runs both instances in one process
    if (gen && atoi(gen) != 0) return US_Q_RUNTIME;
    const uint32_t host_only = PBRT_USQ_NO_FIRST_TABLES | PBRT_USQ_NO_FUSED_BOUNCES;  // decided on the host: tables null, a.fuse 0
    const uint32_t q = a.p.quirks & ~host_only;
    return (q == PBRT_USQ_REFERENCE || q == (PBRT_USQ_REFERENCE | PBRT_USQ_NO_CARRIER)) ? q : US_Q_RUNTIME;
}
template <bool FIRST, int ACCEL, bool EMIT>
static void launch_us_instance(const UsArgs &a, uint32_t nseg, uint32_t threads, size_t lds, hipStream_t st) {
    const uint32_t q = us_kernel_quirks(a);
    const int tab = !FIRST ? -1 : (a.first_hit && a.first_rx) ? 1 : (!a.first_hit && !a.first_rx) ? 0 : -1;
#define US_LAUNCH(QQ, TT) hipLaunchKernelGGL((k_us_bounce<FIRST, ACCEL, EMIT, QQ, TT>), dim3(nseg), dim3(threads), lds, st, a)
    if constexpr (ACCEL == ACCEL_K_BRUTE) {  // the kernels of BASELINE config 3 (both readings) and of the reference's own loop
        if (q != US_Q_RUNTIME && (tab >= 0 || !FIRST)) {
            const bool carrier = q == PBRT_USQ_REFERENCE;
            if constexpr (!FIRST) {
                if (carrier) US_LAUNCH(PBRT_USQ_REFERENCE, -1); else US_LAUNCH(PBRT_USQ_REFERENCE | PBRT_USQ_NO_CARRIER, -1);
            } else if constexpr (EMIT) {  // (emitter rays: never any tables)
                if (carrier) US_LAUNCH(PBRT_USQ_REFERENCE, 0); else US_LAUNCH(PBRT_USQ_REFERENCE | PBRT_USQ_NO_CARRIER, 0);
            } else {
                if (tab == 1) {
                    if (carrier) US_LAUNCH(PBRT_USQ_REFERENCE, 1); else US_LAUNCH(PBRT_USQ_REFERENCE | PBRT_USQ_NO_CARRIER, 1);
                } else {
                    if (carrier) US_LAUNCH(PBRT_USQ_REFERENCE, 0); else US_LAUNCH(PBRT_USQ_REFERENCE | PBRT_USQ_NO_CARRIER, 0);
                }
            }
            return;
        }
    }
    US_LAUNCH(US_Q_RUNTIME, -1);
#undef US_LAUNCH
}

template <bool FIRST>
static int launch_us(pbrt_scene *s, const UsArgs &a, uint32_t nseg) {
    hipStream_t st = s->ctx->stream;
    if (a.p.primary == PBRT_US_PRIMARY_EMITTER) {  // primary rays from CustomEmitter.sample_ray, echoes times the ray's weight
        switch (s->accel_kernel) {
            case ACCEL_K_BRUTE:
                launch_us_instance<FIRST, ACCEL_K_BRUTE, true>(a, nseg, SEG_BRUTE, 0, st);
                return PBRT_OK;
            case ACCEL_K_BRUTE_BIG:
                launch_us_instance<FIRST, ACCEL_K_BRUTE_BIG, true>(a, nseg, SEG_BRUTE, 0, st);
                return PBRT_OK;
            default:
                return s->ctx->fail(PBRT_E_UNSUPPORTED, "launch_us: emitter primary rays on BVH scenes run as streams (us_wf_pass)");
        }
    }
    switch (s->accel_kernel) {
        case ACCEL_K_BRUTE:
            launch_us_instance<FIRST, ACCEL_K_BRUTE, false>(a, nseg, SEG_BRUTE, 0, st);
            break;
        case ACCEL_K_BRUTE_BIG:
            launch_us_instance<FIRST, ACCEL_K_BRUTE_BIG, false>(a, nseg, SEG_BRUTE, 0, st);
            break;
#ifdef PBRT_DIAG  // the fused ultrasound bounce on BVH scenes (PBRT_US_FUSED_BVH=1): diagnostic build only
        case ACCEL_K_BVH_GLOBAL:
            launch_us_instance<FIRST, ACCEL_K_BVH_GLOBAL, false>(a, nseg, SEG_BVH, 0, st);
            break;
        default:
            launch_us_instance<FIRST, ACCEL_K_BVH_LDS, false>(a, nseg, SEG_BVH, s->lds_bytes, st);
            break;
#else
        default:  // BVH scenes run k_trace / k_us_shade (us_wf_pass)
            return s->ctx->fail(PBRT_E_UNSUPPORTED, "launch_us: no fused ultrasound bounce for accelerator %d in this build", s->accel_kernel);
#endif
    }
    return PBRT_OK;
}

// BVH scenes: the bounces of one ultrasound pass as k_trace / k_us_shade streams (kernels_us_wavefront.h).  a: the pass's UsArgs
// (cap, n_paths, ppr_pass, path_first, tables set).  One launch at depth 0 with the first-bounce tables (else k_us_init_wf +
// k_trace + k_us_shade), two per later bounce, and a flush for the occlusion rays of the last one.
static int us_wf_pass(pbrt_scene *s, UsArgs a, const WfBufs &b, const WfPlan &p, uint32_t nreg, uint32_t *launches) {
    pbrt_ctx *c = s->ctx;
    hipStream_t st = c->stream;
    WfArgs t{};
    t.sc = s->ds;
    t.cap = a.cap;
    t.n_paths = a.n_paths;
    t.key_mode = 0;
    t.vis_q = US_WF_VIS_Q;
    t.hit_id = b.hit_id;
    t.lds_bytes = s->accel_kernel == ACCEL_K_BVH_LDS ? s->lds_bytes : 0u;
    t.stk_rows = p.rows;
    t.stk_shift = 0;
    while ((1u << t.stk_shift) < p.threads) ++t.stk_shift;
    t.region0 = 0;
    t.n_regions = nreg;
    if (!(t.guard = wf_guard(c))) return PBRT_E_NOMEM;
    UsWfArgs w{};
    w.hit_id = b.hit_id;
    w.region0 = 0;
    w.n_regions = nreg;
    w.guard = t.guard;
    float4 *in = b.stA, *out = b.stB, *shi = b.shA, *sho = b.shB;
    uint32_t *sin = b.segA, *sout = b.segB, *ni = b.nshA, *no = b.nshB;
    auto trace = [&](uint32_t depth, bool have_shadows) {
        t.depth = depth;
        t.st_in = in;
        t.shd_in = shi;
        t.seg_in = sin;
        t.nsh_in = have_shadows ? ni : nullptr;
        wf_launch_trace(s, t, p, wf_trace_grid(nreg, depth >= 2 ? p.grid_deep : p.grid_mult, (uint32_t)c->n_cu), false, st);
        ++*launches;
    };
    auto shade = [&](uint32_t depth, bool tab, bool have_shadows) {
        a.depth = depth;
        w.u = a;
        w.st_in = in;
        w.st_out = out;
        w.shd_in = shi;
        w.shd_out = sho;
        w.seg_in = sin;
        w.seg_out = sout;
        w.nsh_in = have_shadows ? ni : nullptr;
        w.nsh_out = no;
        if (tab)
            hipLaunchKernelGGL(k_us_shade<true>, dim3(nreg), dim3(WF_SHADE_THREADS), 0, st, w);
        else
            hipLaunchKernelGGL(k_us_shade<false>, dim3(nreg), dim3(WF_SHADE_THREADS), 0, st, w);
        ++*launches;
    };
    auto flip = [&]() {
        std::swap(in, out);
        std::swap(shi, sho);
        std::swap(sin, sout);
        std::swap(ni, no);
    };
    const bool tab = a.first_hit != nullptr && a.first_rx != nullptr;
    if (tab) {
        shade(0, true, false);
    } else {
        hipLaunchKernelGGL(k_us_init_wf, dim3(div_up(std::max(a.n_paths, nreg), 256)), dim3(256), 0, st, a, in, sin, nreg);
        trace(0, false);
        shade(0, false, false);
    }
    flip();
    HIPCHK(c, hipGetLastError());
    for (uint32_t depth = 1; depth < a.p.max_depth; ++depth) {
        trace(depth, true);
        shade(depth, false, true);
        flip();
        HIPCHK(c, hipGetLastError());
    }
    // the occlusion rays of the last bounce (every path has ended: records of ended paths only), and their echoes
    trace(a.p.max_depth, true);
    shade(a.p.max_depth, false, true);
    HIPCHK(c, hipGetLastError());
    return PBRT_OK;
}

extern "C" {

static int us_impl(pbrt_scene *s, const pbrt_us_params *p, uint32_t seed, uint32_t ppr, uint32_t path_offset,
                   uint32_t norm_paths, float *d_channel, float *tx_host, bool wait = true) {
    pbrt_ctx *c = s->ctx;
    if (c->recording) {  // recorded, not run: the queueing form only, one acquisition per recording
        if (wait || c->pend.active) return c->fail(PBRT_E_INVALID, "recording: one pbrt_us_acquire_queue_dev per recording, and no call that waits");
    } else if (int rcs = ctx_settle(c)) {
        return rcs;
    }
    const bool timed = !c->recording;  // (event pairs recorded into a graph cannot be read back)
    NEED(c, p && d_channel);
    NEED(c, p->n_angles > 0 && p->n_angles <= PBRT_US_MAX_ANGLES && p->n_elements > 0 && p->time_samples > 0);
    NEED(c, (uint64_t)p->n_angles * p->n_elements * p->time_samples < 0xffffffffull);  // channel index is 32-bit (echo bins)
    NEED(c, p->max_depth > 0 && ppr > 0 && p->sound_speed > 0 && p->fs > 0);
    NEED(c, p->max_depth < 0x40000000u);  // RNG block = bounce index; bit 30 marks a path's second block of a bounce, bit 31 the emitter's
    NEED(c, p->primary <= PBRT_US_PRIMARY_EMITTER);
    const bool emit = p->primary == PBRT_US_PRIMARY_EMITTER;
    if (emit) {  // CustomEmitter as the source of the primary rays: its elements are the acquisition's (include/pbrt_hip.h)
        const pbrt_us_emitter &E = p->emitter;
        NEED(c, E.number_of_elements == p->n_elements && E.number_of_rays_per_element > 0 && E.speed_of_sound > 0.0f);
        NEED(c, std::isfinite(E.pitch) && std::isfinite(E.element_width) && std::isfinite(E.element_height) && std::isfinite(E.radius));
        NEED(c, std::isfinite(E.steering_angle_min) && std::isfinite(E.steering_angle_max) && std::isfinite(E.opening_angle));
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = set_lds_attr(s);
    if (rc) return rc;
    const uint32_t NA = p->n_angles, NE = p->n_elements, T = p->time_samples;
    const uint32_t n_rays = NA * NE;
    std::vector<float> tx(n_rays), dir0(3 * NA), ex(NE);
    pbrt_us_tx_delays(p, tx.data());
    if (tx_host) std::memcpy(tx_host, tx.data(), tx.size() * 4);
    const float *M = p->sensor_to_world;
    auto xfv = [&](float x, float y, float z, float *o) {
        o[0] = host_fma(M[0], x, host_fma(M[1], y, M[2] * z));
        o[1] = host_fma(M[4], x, host_fma(M[5], y, M[6] * z));
        o[2] = host_fma(M[8], x, host_fma(M[9], y, M[10] * z));
    };
    auto nrm = [&](float *v) {
        float inv = 1.0f / std::sqrt(host_fma(v[0], v[0], host_fma(v[1], v[1], v[2] * v[2])));
        v[0] *= inv;
        v[1] *= inv;
        v[2] *= inv;
    };
    for (uint32_t a = 0; a < NA; ++a) {  // CustomIntegrator.py:265,271,273
        float ar = (float)((double)p->angles_deg[a] * (M_PI / 180.0));
        xfv(sinf(ar), 0.0f, cosf(ar), &dir0[3 * a]);
        nrm(&dir0[3 * a]);
    }
    for (uint32_t e = 0; e < NE; ++e)
        ex[e] = (float)((double)p->pitch * ((double)(float)e - ((double)NE - 1.0) / 2.0));  // :248
#ifndef US_PASS_PATHS
// paths in flight per pass.  Config 3 (268 M paths), one launch per bounce: 8 / 16 / 32 / 64 Mi -> 13.1 / 12.7 / 13.0 /
// 13.2 ms; with all bounces of a pass in one launch the survivors are re-read while still cached and smaller passes
// win: 2 / 4 / 6 / 8 / 12 / 16 / 32 / 64 Mi -> 13.1 / 9.8 / 8.4 / 8.1 / 8.2 / 8.5 / 9.0 / 8.7 ms.  Later in round 2 (Mitsuba's
// shading frame: 1.96 segments per path; three workgroups per CU): 4 / 8 / 16 / 32 Mi -> 13.26 / 11.89 / 11.61 / 12.20 ms
#define US_PASS_PATHS (16u << 20)
#endif
    // BVH scenes (tessellated phantoms, meshes) run as k_trace / k_us_shade streams (us_wf_pass); the diagnostic build keeps the
    // fused bounce behind PBRT_US_FUSED_BVH=1 for the A/B
    const bool bvh_scene = s->accel_kernel == ACCEL_K_BVH_GLOBAL || s->accel_kernel == ACCEL_K_BVH_LDS;
    bool streams = bvh_scene;
#ifdef PBRT_DIAG
    if (const char *e = getenv("PBRT_US_FUSED_BVH")) streams = streams && atoi(e) == 0;
#endif
    // paths in flight per pass.  Streams: 2 x 10 launches per pass whatever is still alive, so passes as large as the radiance
    // streams take (wf_default_pass_paths; the shared trace / shade workspace)
    uint64_t pass_paths = streams ? wf_default_pass_paths(c, 1u << 20) : US_PASS_PATHS;
    if (!streams && c->ws_limit)  // the pass buffers must fit the context's workspace limit
        while (pass_paths > (1u << 20) && (double)pass_paths * (double)(2 * N_STATE * 4) > (double)c->ws_limit - 64e6) pass_paths /= 2;
    // The free-memory figure behind the default is a snapshot (torch or another process may allocate between the query and the
    // hipMalloc): a failed allocation gives the pass buffers back, halves the pass and tries again, like render_impl; the
    // echoes do not depend on the pass size (global path keys).  What did fit is released when even the smallest pass does not.
    const uint64_t US_MIN_PASS = 1u << 20;
    const uint32_t REGION = streams ? WF_REGION : us_region_segs(s->accel_kernel, emit) * seg_threads(s->accel_kernel);
    uint32_t ppr_pass = 1, cap = 0, nseg = 0;
    WfBufs wfb{};
    WfPlan wfp;
    float *stA = nullptr, *stB = nullptr;
    uint32_t *segA = nullptr, *segB = nullptr;
    for (;;) {
        ppr_pass = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(ppr, pass_paths / n_rays));
        NEED(c, (uint64_t)n_rays * ppr_pass < 0xfffffc00ull);
        cap = div_up((uint64_t)n_rays * ppr_pass, REGION) * REGION;
        nseg = cap / REGION;
        NEED(c, streams || (uint64_t)cap * US_N_STATE * 4 < 0xffffffffull);  // the state tiles are addressed through 32-bit buffer offsets
        NEED(c, !streams || cap < WF_DEAD);
        bool ok;
        if (streams) {
            ok = wf_alloc(c, cap, nseg, &wfb);
        } else {
            stA = (float *)c->buf("stateA", (size_t)cap * N_STATE * 4);
            stB = stA ? (float *)c->buf("stateB", (size_t)cap * N_STATE * 4) : nullptr;
            ok = stA && stB;
        }
        if (ok) break;
        release_pass_buffers(c);
        if (ppr_pass <= 1 || pass_paths <= US_MIN_PASS) return PBRT_E_NOMEM;  // (the message of the failed request stands)
        pass_paths = std::max<uint64_t>(US_MIN_PASS, std::min<uint64_t>(pass_paths, (uint64_t)n_rays * ppr_pass) / 2);
    }
    if (streams) {
        wfp = wf_plan(s);
        wfp.packet = false;
        if ((rc = wf_set_attr(s, wfp)) != 0) return rc;
    }
    // live counters / statistics rows: per region, per wave of a region for the BVH kernels
    const uint32_t n_own = nseg * (streams ? WF_SHADE_THREADS / 64u : us_owners_per_region(s->accel_kernel));
    if (!streams) {
        segA = (uint32_t *)c->buf("segA", (size_t)n_own * 4);
        segB = (uint32_t *)c->buf("segB", (size_t)n_own * 4);
    }
    // the counter rows: summed into the pinned page and zeroed again by k_us_reduce_stats at the end of the call
    const size_t segstats_bytes = (size_t)(2 + MAX_DEPTH_STATS) * n_own * 8;
    unsigned long long *dstats = (unsigned long long *)c->buf("us_stats", segstats_bytes);
    if (!dstats) return PBRT_E_NOMEM;
    unsigned long long *segstats = dstats;
    float *tabs = (float *)c->buf("us_tables", ((size_t)n_rays + 3 * NA + NE) * 4);
    if ((!streams && (!segA || !segB)) || !dstats || !tabs) return PBRT_E_NOMEM;
    hipStream_t st = c->stream;
    float *d_tx = tabs, *d_dir = tabs + n_rays, *d_ex = d_dir + 3 * NA;
    // emitter rays are drawn inside the first-bounce instance (k_us_bounce<true, ., EMIT>: 17.1 against 18.4 ms since the echo table
    // grew); PBRT_US_EMIT_FUSED=0 (A/B, test) writes them into the path state first (k_us_emit_init) and walks every bounce with the
    // later-bounce instance.  (Two default bench runs of this form showed steps of 38 - 43 ms; host stalls of that size hit the
    // two-kernel form as well and went away when bench.py took Python's cyclic collector out of its timed steps: three runs of
    // either form clean since.)
    const char *e_fused = getenv("PBRT_US_EMIT_FUSED");
    const bool emit_fused = !(e_fused && atoi(e_fused) == 0);
    // the three small tables in one host image; uploaded only when they differ from what the device copy already holds
    {
        std::vector<float> img((size_t)n_rays + 3 * NA + NE);
        if (emit)  // the ray's own emission time rides in its time of flight (CustomEmmitter.py:93-94); t0 of :329 is 0
            std::fill(img.begin(), img.begin() + n_rays, 0.0f);
        else
            std::copy(tx.begin(), tx.end(), img.begin());
        std::copy(dir0.begin(), dir0.end(), img.begin() + n_rays);
        std::copy(ex.begin(), ex.end(), img.begin() + n_rays + 3 * NA);
        if (c->us_tab_dev != (const void *)tabs || c->us_tab_host != img) {
            if (c->recording) return c->fail(PBRT_E_INVALID, "recording: the acquisition's tables are not on the device yet -- run the chain once before recording it");
            ++c->ws_epoch;  // (a finished recording was made with the tables that are replaced now)
            HIPCHK(c, hipMemcpyAsync(tabs, img.data(), img.size() * 4, hipMemcpyHostToDevice, st));
            c->us_tab_host.swap(img);
            c->us_tab_dev = tabs;
        }
    }
    const size_t nchan = (size_t)n_rays * T;
    HIPCHK(c, hipMemsetAsync(d_channel, 0, nchan * 4, st));
    // (the rows are clean if the last acquisition's reduction has swept exactly this buffer; anything else -- a fresh or resized buffer,
    // a call that failed half-way -- gets the fill)
    if (c->us_rows_clean != (const void *)dstats || c->us_rows_clean_bytes != segstats_bytes || c->us_rows_epoch != c->ws_epoch) {
        if (c->recording) return c->fail(PBRT_E_INVALID, "recording: the acquisition's counters are not clean yet -- run the chain once before recording it");
        HIPCHK(c, hipMemsetAsync(dstats, 0, segstats_bytes, st));
    }
    c->us_rows_clean = nullptr;
    if (timed) HIPCHK(c, hipEventRecord(c->ev0, st));
    UsArgs a{};
    a.sc = s->ds;
    a.p = *p;
    a.stats = segstats;
    a.stat_stride = n_own;
    a.channel = d_channel;
    a.tx = d_tx;
    a.dir0 = d_dir;
    a.elem_x = d_ex;
    float tn[3];
    xfv(0.0f, 0.0f, 1.0f, tn);
    nrm(tn);
    a.tn[0] = tn[0];
    a.tn[1] = tn[1];
    a.tn[2] = tn[2];
    a.am = p->main_beam_angle * (K_PI / 180.0f);
    a.ac = p->cutoff_angle * (K_PI / 180.0f);
    a.cos_min = cosf(a.ac);                                                           // :370
    a.katt = (float)(-(double)p->attenuation * (double)p->frequency * 1e-6);          // :328
    a.two_pi_f = (float)(2.0 * M_PI * (double)p->frequency);                          // :330
    a.inv_c = 1.0f / p->sound_speed;
    a.cap = cap;
    a.seed = seed;
    a.lds_bytes = s->lds_bytes;
    a.fuse = (p->quirks & PBRT_USQ_NO_FUSED_BOUNCES) ? 0u : 1u;
    a.div_ne = make_fastdiv(NE);
    for (uint32_t n : {0u, 1u, NE - 1, NE, NE + 1, n_rays - 1, n_rays}) NEED(c, udiv_fast_host(n, a.div_ne) == n / NE);
    // first-bounce tables (kernels_us.h k_us_first): worth it once a ray has more paths than receive elements
    if (ppr >= NE && !(p->quirks & PBRT_USQ_NO_FIRST_TABLES) && !emit) {  // (emitter rays: every path has its own origin)
        float4 *fh = (float4 *)c->buf("us_first_hit", (size_t)n_rays * 16);
        float4 *fv = (float4 *)c->buf("us_first_rx", (size_t)n_rays * NE * 16);
        if (!fh || !fv) return PBRT_E_NOMEM;
        const dim3 g(div_up((uint64_t)n_rays * NE, 256)), b(256);
        if (s->accel_kernel == ACCEL_K_BRUTE || s->accel_kernel == ACCEL_K_BRUTE_BIG)
            hipLaunchKernelGGL(k_us_first<ACCEL_K_BRUTE_BIG>, g, b, 0, st, a, n_rays, fh, fv);
        else
            hipLaunchKernelGGL(k_us_first<ACCEL_K_BVH_GLOBAL>, g, b, 0, st, a, n_rays, fh, fv);
        HIPCHK(c, hipGetLastError());
        a.first_hit = fh;
        a.first_rx = fv;
    }
    size_t n_ev = 0;
    hipEvent_t pass_e1 = nullptr;
    uint32_t passes = 0, launches = 0;
    for (uint32_t k0 = 0; k0 < ppr; k0 += ppr_pass, ++passes) {
        const uint32_t kc = std::min(ppr_pass, ppr - k0);
        a.ppr_pass = kc;
        a.div_ppr = make_fastdiv(kc);
        for (uint32_t n : {0u, 1u, kc - 1, kc, kc + 1, n_rays * kc - 1, n_rays * kc, 0xfffffbffu})
            NEED(c, udiv_fast_host(n, a.div_ppr) == n / kc);
        a.path_first = path_offset + k0;
        a.n_paths = n_rays * kc;
        const uint32_t nseg_pass = div_up(a.n_paths, REGION);
        a.blk_mul = 0;
        const char *e_perm = getenv("PBRT_US_EMIT_PERMUTE");  // A/B and test: 0 keeps workgroup b on region b (read per call)
        const char *e_perm_all = getenv("PBRT_US_PERMUTE_ALL");  // A/B: the integrator's own rays as well
        if ((emit || (e_perm_all && atoi(e_perm_all) != 0)) && !streams && nseg_pass > 2u * NA && !(e_perm && atoi(e_perm) == 0)) {
            uint32_t m = (nseg_pass / NA) | 1u;
            if (e_perm && atoi(e_perm) > 1) m = (uint32_t)atoi(e_perm) | 1u;  // (A/B: another stride)
            while (std::gcd(m, nseg_pass) != 1u) m += 2u;
            a.blk_mul = m % nseg_pass;
        }
        if (streams) {
            pass_e1 = c->event(n_ev + 1);
            hipEvent_t e0 = c->event(n_ev);
            if (!e0 || !pass_e1) return c->fail(PBRT_E_DEVICE, "hipEventCreate failed");
            n_ev += 2;
            if (timed) HIPCHK(c, hipEventRecord(e0, st));
            if ((rc = us_wf_pass(s, a, wfb, wfp, nseg_pass, &launches)) != 0) return rc;
            if (timed) HIPCHK(c, hipEventRecord(pass_e1, st));
            continue;
        }
        float *in = stA, *out = stB;
        uint32_t *sin = segA, *sout = segB;
        for (uint32_t depth = 0; depth < p->max_depth; ++depth) {
            a.depth = depth;
            a.in = in;
            a.out = out;
            a.seg_in = sin;
            a.seg_out = sout;
            // ONE event pair per pass around its bounce launches (a pair per launch costs ~8 us of queue bubbles each)
            if (depth == 0) {
                pass_e1 = c->event(n_ev + 1);
                hipEvent_t e0 = c->event(n_ev);
                if (!e0 || !pass_e1) return c->fail(PBRT_E_DEVICE, "hipEventCreate failed");
                n_ev += 2;
                if (timed) HIPCHK(c, hipEventRecord(e0, st));
            }
            bool first_kernel = depth == 0;
            if (depth == 0 && emit && !emit_fused) {
                // emitter rays: the primary rays into the state (k_us_emit_init writes a.out / a.seg_out), then the later-bounce
                // instance from depth 0
                hipLaunchKernelGGL(k_us_emit_init, dim3(div_up(std::max(a.n_paths, nseg_pass), 256)), dim3(256), 0, st, a, REGION, nseg_pass);
                std::swap(in, out);
                std::swap(sin, sout);
                a.in = in;
                a.out = out;
                a.seg_in = sin;
                a.seg_out = sout;
                first_kernel = false;
            }
            if ((rc = first_kernel ? launch_us<true>(s, a, nseg_pass) : launch_us<false>(s, a, nseg_pass)) != 0) return rc;
            HIPCHK(c, hipGetLastError());
            ++launches;
            if (a.fuse) break;  // that launch walked every bounce (kernels_us.h)
            std::swap(in, out);
            std::swap(sin, sout);
        }
        if (timed) HIPCHK(c, hipEventRecord(pass_e1, st));
    }
    const float inv_norm = 1.0f / (float)(norm_paths ? norm_paths : 1);
    // (one path per ray, the reference's own setting (USMain.py:36): the factor is exactly 1 and the pass over the buffer changes no bit)
    if (inv_norm != 1.0f) hipLaunchKernelGGL(k_scale, dim3(div_up(nchan, 256)), dim3(256), 0, st, d_channel, nchan, inv_norm);
    HIPCHK(c, hipGetLastError());
    if (timed) HIPCHK(c, hipEventRecord(c->ev1, st));
    // counters into the context's pinned page (partial sums, written by the kernel; the rows are zero again behind it), guard words by
    // a queued copy; both are read by us_finish once the stream has drained
    hipLaunchKernelGGL(k_us_reduce_stats, dim3(2 + MAX_DEPTH_STATS, REDUCE_SLICES), dim3(256), 0, st, segstats, n_own, (size_t)n_own, c->pin_stats());
    HIPCHK(c, hipGetLastError());
    c->us_rows_clean = dstats;
    c->us_rows_clean_bytes = segstats_bytes;
    c->us_rows_epoch = c->ws_epoch;
    if (streams && (rc = wf_guard_fetch(c, c->pin_guard())) != 0) return rc;
    pbrt_ctx::PendingAcq &P = c->pend;
    P.active = true;
    P.timed = timed;
    P.scaled = inv_norm != 1.0f;
    P.streams = streams;
    P.tab0 = a.first_hit != nullptr;
    P.n_ev = n_ev;
    P.passes = passes;
    P.launches = launches;
    P.samples = (uint64_t)n_rays * ppr;
    P.nchan = nchan;
    return wait ? us_finish(c) : PBRT_OK;
}

}  // extern "C"

static int us_finish(pbrt_ctx *c) {
    pbrt_ctx::PendingAcq P = c->pend;
    c->pend.active = false;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIPCHK(c, hipStreamSynchronize(st));
    int rc;
    if (P.streams && (rc = wf_check_guard(c, c->pin_guard())) != 0) return rc;
    unsigned long long hstats[2 + MAX_DEPTH_STATS];  // the slices of a row, added here
    for (uint32_t r = 0; r < 2 + MAX_DEPTH_STATS; ++r) {
        unsigned long long t = 0;
        for (uint32_t sl = 0; sl < REDUCE_SLICES; ++sl) t += c->pin_stats()[(size_t)r * REDUCE_SLICES + sl];
        hstats[r] = t;
    }
    float ms = 0.0f;
    if (P.timed) HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    double bounce_ms = 0.0;
    for (size_t i = 0; P.timed && i + 1 < P.n_ev; i += 2) {
        float t = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&t, c->ev_pool[i], c->ev_pool[i + 1]));
        bounce_ms += t;
    }
    pbrt_stats &S = c->stats;
    S = pbrt_stats{};
    S.samples = P.samples;
    S.segments = hstats[0];
    S.shadow_rays = hstats[1];
    S.kernel_ms = ms;
    S.bounce_ms = bounce_ms;
    S.bounce_launches = P.launches;
    S.passes = P.passes;
    for (int d = 0; d < 16; ++d) S.live[d] = hstats[2 + d];
    uint64_t bb = 0;
    for (uint32_t d = 0; d < MAX_DEPTH_STATS; ++d) {
        uint64_t in = hstats[2 + d], next = d + 1 < MAX_DEPTH_STATS ? hstats[2 + d + 1] : 0;
        if (d > 0) bb += in * (N_USTATE * 4);
        bb += next * (N_USTATE * 4);
    }
    bb += hstats[0] * 8;  // one f32 atomic (read-modify-write) per shaded segment, upper bound
    if (P.streams) {
        // k_trace + k_us_shade (kernels_us_wavefront.h), the bytes the algorithm needs: per ray that is traced 32 B read + 4 B hit
        // index written and read back; a path's pending echo (32 B) is read once,
        // the rest of its state (32 B) if it hit; 64 B per survivor; an occlusion ray is 32 B written, 32 B read, 4 B answered.
        // (With the first-bounce tables depth 0 traces and reads nothing.)  Occlusion rays: one per shaded segment at most.
        const bool tab0 = P.tab0;
        uint64_t tr = 0, sh = 0;
        for (uint32_t d = 0; d < MAX_DEPTH_STATS; ++d) {
            const uint64_t in = hstats[2 + d], next = d + 1 < MAX_DEPTH_STATS ? hstats[2 + d + 1] : 0;
            if (!in) break;
            if (d > 0 || !tab0) {
                tr += in * (32 + 4);
                sh += in * (4 + 32);
            }
            sh += next * 64;
        }
        const uint64_t seg = hstats[0], seg_traced = tab0 ? seg - std::min<uint64_t>(seg, hstats[2]) : seg;  // (every path of depth 0 hits or none of its ray does)
        tr += seg_traced * 36;
        sh += seg_traced * (32 + 32) + seg * 8;
        bb = tr + sh;
        S.trace_model_bytes = tr;
    }
    S.bounce_model_bytes = bb;
    S.model_bytes = bb + P.nchan * (P.scaled ? 12 : 4);  // clear (+ scale pass) over the channel buffer
    S.workspace_bytes = c->ws_total();
    return PBRT_OK;
}

extern "C" {

int pbrt_us_acquire_dev(pbrt_scene *s, const pbrt_us_params *p, uint32_t seed, uint32_t ppr, uint32_t path_offset,
                        uint32_t norm_paths, void *d_channel, float *tx) {
    if (!s) return PBRT_E_INVALID;
    ++s->ctx->call_seq;
    return us_impl(s, p, seed, ppr, path_offset, norm_paths, (float *)d_channel, tx);
}

int pbrt_us_acquire_queue_dev(pbrt_scene *s, const pbrt_us_params *p, uint32_t seed, uint32_t ppr, uint32_t path_offset,
                              uint32_t norm_paths, void *d_channel, float *tx) {
    if (!s) return PBRT_E_INVALID;
    ++s->ctx->call_seq;
    return us_impl(s, p, seed, ppr, path_offset, norm_paths, (float *)d_channel, tx, false);
}

int pbrt_us_acquire(pbrt_scene *s, const pbrt_us_params *p, uint32_t seed, uint32_t ppr, uint32_t path_offset,
                    uint32_t norm_paths, float *channel, float *tx) {
    if (!s) return PBRT_E_INVALID;
    pbrt_ctx *c = s->ctx;
    NEED(c, p && channel);
    HIPCHK(c, hipSetDevice(c->device));
    ++c->call_seq;
    const size_t n = (size_t)p->n_angles * p->n_elements * p->time_samples;
    void *d = c->buf("us_channel", n * 4);
    if (!d) return PBRT_E_NOMEM;
    int rc = us_impl(s, p, seed, ppr, path_offset, norm_paths, (float *)d, tx);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(channel, d, n * 4, hipMemcpyDeviceToHost));
    return PBRT_OK;
}

// ------------------------------------------------------------------------------------------------
// leaf operators: stage host SoA through the workspace, one kernel, copy back
// ------------------------------------------------------------------------------------------------
}  // extern "C"

struct Stage {
    pbrt_ctx *c;
    char *base = nullptr;
    size_t off = 0, cap = 0;
    int rc = PBRT_OK;
    Stage(pbrt_ctx *ctx, const char *name, size_t bytes) : c(ctx) {
        ++c->call_seq;
        base = (char *)c->buf(name, bytes + 1024);
        cap = bytes + 1024;
        if (!base) rc = PBRT_E_NOMEM;
    }
    template <typename T>
    T *in(const T *h, size_t n) {
        T *d = reinterpret_cast<T *>(base + off);
        off += (n * sizeof(T) + 15) & ~size_t(15);
        if (h && rc == PBRT_OK && hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, c->stream) != hipSuccess)
            rc = c->fail(PBRT_E_DEVICE, "leaf upload failed");
        return h ? d : nullptr;
    }
    template <typename T>
    T *out(size_t n) {
        T *d = reinterpret_cast<T *>(base + off);
        off += (n * sizeof(T) + 15) & ~size_t(15);
        return d;
    }
    template <typename T>
    void back(T *h, const T *d, size_t n) {
        if (rc == PBRT_OK && hipMemcpyAsync(h, d, n * sizeof(T), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
            rc = c->fail(PBRT_E_DEVICE, "leaf download failed");
    }
    int finish() {
        if (rc != PBRT_OK) return rc;
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return c->fail(PBRT_E_DEVICE, "leaf kernel: %s", hipGetErrorString(e));
        return PBRT_OK;
    }
};

extern "C" {

#define LEAF_BEGIN(ctxp, total_bytes)          \
    pbrt_ctx *c = (ctxp);                      \
    if (!c) return PBRT_E_INVALID;             \
    if (n == 0) return PBRT_OK;                \
    NOT_RECORDING(c);                          \
    HIPCHK(c, hipSetDevice(c->device));        \
    Stage S(c, "leaf_io", (total_bytes));      \
    if (S.rc) return S.rc;                     \
    const dim3 grid(div_up(n, 256)), block(256); \
    hipStream_t st = c->stream;

int pbrt_ray_intersect(pbrt_scene *s, uint32_t n, const float *o, const float *d, const float *tmax, float *t,
                       uint32_t *prim, float *u, float *v) {
    if (!s) return PBRT_E_INVALID;
    NEED(s->ctx, o && d && tmax && t && prim && u && v);
    LEAF_BEGIN(s->ctx, (size_t)n * 4 * 12);
    float *dO = S.in(o, 3 * (size_t)n), *dD = S.in(d, 3 * (size_t)n), *dT = S.in(tmax, n);
    float *rt = S.out<float>(n), *ru = S.out<float>(n), *rv = S.out<float>(n);
    uint32_t *rp = S.out<uint32_t>(n);
    if (s->accel_kernel == ACCEL_K_BRUTE || s->accel_kernel == ACCEL_K_BRUTE_BIG)
        hipLaunchKernelGGL(k_ray_intersect<ACCEL_K_BRUTE_BIG>, grid, block, 0, st, s->ds, n, dO, dD, dT, rt, rp, ru, rv);
    else
        hipLaunchKernelGGL(k_ray_intersect<ACCEL_K_BVH_GLOBAL>, grid, block, 0, st, s->ds, n, dO, dD, dT, rt, rp, ru, rv);
    S.back(t, rt, n);
    S.back(prim, rp, n);
    S.back(u, ru, n);
    S.back(v, rv, n);
    return S.finish();
}

int pbrt_ray_test(pbrt_scene *s, uint32_t n, const float *o, const float *d, const float *tmax, uint8_t *hit) {
    if (!s) return PBRT_E_INVALID;
    NEED(s->ctx, o && d && tmax && hit);
    LEAF_BEGIN(s->ctx, (size_t)n * 4 * 9);
    float *dO = S.in(o, 3 * (size_t)n), *dD = S.in(d, 3 * (size_t)n), *dT = S.in(tmax, n);
    uint8_t *rh = S.out<uint8_t>(n);
    if (s->accel_kernel == ACCEL_K_BRUTE || s->accel_kernel == ACCEL_K_BRUTE_BIG)
        hipLaunchKernelGGL(k_ray_test<ACCEL_K_BRUTE_BIG>, grid, block, 0, st, s->ds, n, dO, dD, dT, rh);
    else
        hipLaunchKernelGGL(k_ray_test<ACCEL_K_BVH_GLOBAL>, grid, block, 0, st, s->ds, n, dO, dD, dT, rh);
    S.back(hit, rh, n);
    return S.finish();
}

int pbrt_bsdf_sample(pbrt_ctx *ctx, const pbrt_material *m, uint32_t quirks, uint32_t n, const float *wi,
                     const float *n_geo, const float *n_sh, const float *sh_s, const float *s1, const float *s2, float *wo,
                     float *pdf, float *weight, uint32_t *sampled) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, m && wi && s1 && s2 && wo && pdf && weight && sampled);
    LEAF_BEGIN(ctx, (size_t)n * 4 * 25);
    float *dwi = S.in(wi, 3 * (size_t)n), *dng = S.in(n_geo, 3 * (size_t)n), *dns = S.in(n_sh, 3 * (size_t)n);
    float *dss = S.in(sh_s, 3 * (size_t)n);
    float *d1 = S.in(s1, n), *d2 = S.in(s2, 2 * (size_t)n);
    float *rwo = S.out<float>(3 * (size_t)n), *rpdf = S.out<float>(n), *rw = S.out<float>(3 * (size_t)n);
    uint32_t *rs = S.out<uint32_t>(n);
    hipLaunchKernelGGL(k_bsdf_sample, grid, block, 0, st, *m, quirks, n, dwi, dng, dns, dss, d1, d2, rwo, rpdf, rw, rs);
    S.back(wo, rwo, 3 * (size_t)n);
    S.back(pdf, rpdf, n);
    S.back(weight, rw, 3 * (size_t)n);
    S.back(sampled, rs, n);
    return S.finish();
}

int pbrt_bsdf_eval_pdf(pbrt_ctx *ctx, const pbrt_material *m, uint32_t n, const float *wi, const float *wo, float *f,
                       float *pdf) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, m && wi && wo && f && pdf);
    LEAF_BEGIN(ctx, (size_t)n * 4 * 12);
    float *dwi = S.in(wi, 3 * (size_t)n), *dwo = S.in(wo, 3 * (size_t)n);
    float *rf = S.out<float>(3 * (size_t)n), *rp = S.out<float>(n);
    hipLaunchKernelGGL(k_bsdf_eval_pdf, grid, block, 0, st, *m, n, dwi, dwo, rf, rp);
    S.back(f, rf, 3 * (size_t)n);
    S.back(pdf, rp, n);
    return S.finish();
}

int pbrt_emitter_sample_direction(pbrt_scene *s, uint32_t n, const float *p, const float *u, float *d, float *dist,
                                  float *pdf, float *weight, float *q, uint32_t *emitter) {
    if (!s) return PBRT_E_INVALID;
    NEED(s->ctx, p && u && d && dist && pdf && weight && q && emitter);
    LEAF_BEGIN(s->ctx, (size_t)n * 4 * 22);
    float *dp = S.in(p, 3 * (size_t)n), *du = S.in(u, 4 * (size_t)n);
    float *rd = S.out<float>(3 * (size_t)n), *rdist = S.out<float>(n), *rpdf = S.out<float>(n);
    float *rw = S.out<float>(3 * (size_t)n), *rq = S.out<float>(3 * (size_t)n);
    uint32_t *re = S.out<uint32_t>(n);
    hipLaunchKernelGGL(k_emitter_sample, grid, block, 0, st, s->ds, n, dp, du, rd, rdist, rpdf, rw, rq, re);
    S.back(d, rd, 3 * (size_t)n);
    S.back(dist, rdist, n);
    S.back(pdf, rpdf, n);
    S.back(weight, rw, 3 * (size_t)n);
    S.back(q, rq, 3 * (size_t)n);
    S.back(emitter, re, n);
    return S.finish();
}

int pbrt_sensor_sample_ray(pbrt_ctx *ctx, const pbrt_camera *cam, uint32_t n, const float *pos, float *o, float *d,
                           float *tmax) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, cam && pos && o && d && tmax);
    LEAF_BEGIN(ctx, (size_t)n * 4 * 10);
    float *dp = S.in(pos, 2 * (size_t)n);
    float *ro = S.out<float>(3 * (size_t)n), *rd = S.out<float>(3 * (size_t)n), *rt = S.out<float>(n);
    hipLaunchKernelGGL(k_sensor_sample_ray, grid, block, 0, st, *cam, n, dp, ro, rd, rt);
    S.back(o, ro, 3 * (size_t)n);
    S.back(d, rd, 3 * (size_t)n);
    S.back(tmax, rt, n);
    return S.finish();
}

int pbrt_us_sensor_sample_ray(pbrt_ctx *ctx, const pbrt_us_sensor *sn, int use_hemisphere_warp, uint32_t n,
                              const float *time, const float *wavelength_sample, const float *position_sample,
                              const float *aperture_sample, float *o, float *d, float *weight) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, sn && time && wavelength_sample && position_sample && aperture_sample && o && d && weight);
    LEAF_BEGIN(ctx, (size_t)n * 4 * 14);
    float *dt = S.in(time, n), *dw = S.in(wavelength_sample, n), *dp = S.in(position_sample, 2 * (size_t)n),
          *da = S.in(aperture_sample, 2 * (size_t)n);
    float *ro = S.out<float>(3 * (size_t)n), *rd = S.out<float>(3 * (size_t)n), *rw = S.out<float>(n);
    hipLaunchKernelGGL(k_us_sensor_sample_ray, grid, block, 0, st, *sn, use_hemisphere_warp, n, dt, dw, dp, da, ro, rd, rw);
    S.back(o, ro, 3 * (size_t)n);
    S.back(d, rd, 3 * (size_t)n);
    S.back(weight, rw, n);
    return S.finish();
}

int pbrt_us_emitter_sample_ray(pbrt_ctx *ctx, const pbrt_us_emitter *e, uint32_t n, const float *time, const float *s1,
                               const float *s2, const float *s3, float *o, float *d, float *ray_time, float *weight,
                               float *pdf_pos) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, e && time && s1 && s2 && s3 && o && d && ray_time && weight && pdf_pos);
    LEAF_BEGIN(ctx, (size_t)n * 4 * 16);
    float *dt = S.in(time, n), *d1 = S.in(s1, n), *d2 = S.in(s2, 2 * (size_t)n), *d3 = S.in(s3, n);
    float *ro = S.out<float>(3 * (size_t)n), *rd = S.out<float>(3 * (size_t)n), *rt = S.out<float>(n),
          *rw = S.out<float>(n), *rp = S.out<float>(n);
    hipLaunchKernelGGL(k_us_emitter_sample_ray, grid, block, 0, st, *e, n, dt, d1, d2, d3, ro, rd, rt, rw, rp);
    S.back(o, ro, 3 * (size_t)n);
    S.back(d, rd, 3 * (size_t)n);
    S.back(ray_time, rt, n);
    S.back(weight, rw, n);
    S.back(pdf_pos, rp, n);
    return S.finish();
}

int pbrt_us_put_data(pbrt_ctx *ctx, const pbrt_us_receiver *r, uint32_t n, const float *ox, const float *time,
                     const float *d, const float *amplitude, float *channel_buffer) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, r && ox && time && d && amplitude && channel_buffer);
    const size_t nb = (size_t)r->number_of_elements * r->time_samples;
    LEAF_BEGIN(ctx, (size_t)n * 4 * 7 + nb * 4);
    float *dx = S.in(ox, n), *dt = S.in(time, n), *dd = S.in(d, 3 * (size_t)n), *da = S.in(amplitude, n);
    float *db = S.in(channel_buffer, nb);
    hipLaunchKernelGGL(k_us_put_data, grid, block, 0, st, *r, n, dx, dt, dd, da, db);
    S.back(channel_buffer, db, nb);
    return S.finish();
}

// ------------------------------------------------------------------------------------------------
// image formation behind the hot path: (pulse ->) beamform -> envelope -> log compression.
// The *_dev entry points queue their kernels on the context's stream and return (ABI 5): the reference's us_render loop
// (USMain.py:92-252, 51 x per run) stays in HBM from the acquisition to the display image.  The host-pointer forms stage
// their arguments and call the same queueing functions.
// ------------------------------------------------------------------------------------------------
}  // extern "C"

// an event pair around one image-formation step when the context profiles (pbrt_ctx_set_profiling); slot = IMG_*
enum { IMG_PULSE = 0, IMG_DAS = 1, IMG_ENV = 2, IMG_LOG = 3, IMG_STEPS = 4 };
struct ImgTimer {
    pbrt_ctx *c;
    int slot;
    ImgTimer(pbrt_ctx *ctx, int s) : c(ctx), slot(s) {
        if (c->profiling && !c->recording && c->img_event(2 * slot)) (void)hipEventRecord(c->img_ev[2 * slot], c->stream);
    }
    ~ImgTimer() {
        if (c->profiling && !c->recording && c->img_event(2 * slot + 1)) {
            (void)hipEventRecord(c->img_ev[2 * slot + 1], c->stream);
            c->img_mask |= 1u << slot;
        }
    }
};

static int das_check(pbrt_ctx *ctx, const pbrt_das_params *p) {
    NEED(ctx, p->n_angles > 0 && p->n_elements > 0 && p->time_samples > 1 && p->fs > 0.0f && p->sound_speed > 0.0f);
    NEED(ctx, p->interpolation <= PBRT_DAS_LINEAR && p->f_number >= 0.0f);
    NEED(ctx, p->nx > 0 && p->nz > 0 && (uint64_t)p->nx * p->nz < 0xffffffffull);
    NEED(ctx, (uint64_t)div_up(p->nx, DAS_TILE) * (div_up(p->nz, DAS_TILE) + DAS_BANDS) * DAS_XCDS < 0x7fffffffull);
    return PBRT_OK;
}
static int das_enqueue(pbrt_ctx *c, const pbrt_das_params *p, const float *dd, const float *dt, const float *de, const float *dx,
                       const float *dz, float *dout, const double *ttx = nullptr) {
    ImgTimer tm(c, IMG_DAS);
    DasGrid g;
    g.ntx = div_up(p->nx, DAS_TILE);
    g.ntz = div_up(p->nz, DAS_TILE);
#ifdef DAS_NO_XCD_BANDS
    g.m = g.ntz;
    const uint32_t blocks = g.ntx * g.ntz;
#else
    // z-tiles of the largest XCD share (bands k and 15 - k, kernels_beamform.h das_tile_of); the grid gives every XCD that many slots
    g.m = 0;
    auto lo = [&](uint32_t band) { return (band * g.ntz + DAS_BANDS - 1u) / DAS_BANDS; };
    for (uint32_t k = 0; k < DAS_XCDS; ++k) g.m = std::max(g.m, (lo(k + 1) - lo(k)) + (lo(DAS_BANDS - k) - lo(DAS_BANDS - 1u - k)));
    const uint32_t blocks = DAS_XCDS * g.ntx * std::max(g.m, 1u);
#endif
    const dim3 grid(blocks), block(64 * DAS_SPLIT);
    if (p->interpolation == PBRT_DAS_NEAREST) {
        if (ttx)
            hipLaunchKernelGGL((k_das_beamform<PBRT_DAS_NEAREST, true>), grid, block, 0, c->stream, *p, g, dd, dt, de, dx, dz, ttx, dout);
        else
            hipLaunchKernelGGL((k_das_beamform<PBRT_DAS_NEAREST, false>), grid, block, 0, c->stream, *p, g, dd, dt, de, dx, dz, ttx, dout);
    } else {
        if (ttx)
            hipLaunchKernelGGL((k_das_beamform<PBRT_DAS_LINEAR, true>), grid, block, 0, c->stream, *p, g, dd, dt, de, dx, dz, ttx, dout);
        else
            hipLaunchKernelGGL((k_das_beamform<PBRT_DAS_LINEAR, false>), grid, block, 0, c->stream, *p, g, dd, dt, de, dx, dz, ttx, dout);
    }
    c->img_das_bytes = ((uint64_t)p->n_angles * p->n_elements * p->time_samples + (uint64_t)p->nx * p->nz) * 4;
    HIPCHK(c, hipGetLastError());
    return PBRT_OK;
}
static int env_enqueue(pbrt_ctx *c, uint32_t nx, uint32_t nz, const float *din, float *dout) {
    const uint32_t np = (nz + 3u) & ~3u, G = 2u * np + 8u;
    // the tap table of this column length: made once, kept while the workspace buffer lives (pbrt_ctx_trim may take it)
    const bool fresh = c->ws.find("env_taps") == c->ws.end() || c->ws["env_taps"].p == nullptr;
    float *taps = (float *)c->buf("env_taps", (size_t)ENV_TAPS_FLOATS * 4);
    if (!taps) return PBRT_E_NOMEM;
    ImgTimer tm(c, IMG_ENV);
    if (fresh || c->env_taps_n != nz) {
        ++c->ws_epoch;  // (a finished recording holds no launch of this kernel: its taps are replaced now)
        hipLaunchKernelGGL(k_hilbert_taps, dim3(div_up(G + 4u * env_even_len(((nz >> 1) + 3u) & ~3u), 256)), dim3(256), 0, c->stream, nz, taps);
        c->env_taps_n = nz;
    }
    // even column lengths: half of the taps are zero, k_hilbert_env_even leaves their multiply-adds out (kernels_beamform.h)
    const uint32_t mp = ((nz >> 1) + 3u) & ~3u;
    const size_t lds_even = (size_t)(2u * mp + 4u * env_even_len(mp)) * 4;  // (its four tap tables: 3.3 x the column; 80 KB at 4096 samples)
    const char *e_gen = getenv("PBRT_ENV_GENERAL");  // A/B and test: every column length through k_hilbert_env (read per call)
    const bool even = (nz & 1u) == 0u && nz >= 8u && lds_even <= (c->lds_limit ? c->lds_limit : 65536u) && !(e_gen && atoi(e_gen) != 0);
    const size_t lds = even ? lds_even : (size_t)(3u * np + 8u) * 4;
    if (lds > c->env_lds_attr) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hilbert_env), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hilbert_env_even), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->env_lds_attr = lds;
    }
    // a thread carries four outputs: whole waves for the quads of a column (638 samples: 160 quads, three waves)
    const uint32_t threads = std::min(256u, div_up(div_up(nz, 4u), 64u) * 64u);
    if (even)
        hipLaunchKernelGGL(k_hilbert_env_even, dim3(nx), dim3(threads), lds, c->stream, nz, din, taps, dout);
    else
        hipLaunchKernelGGL(k_hilbert_env, dim3(nx), dim3(threads), lds, c->stream, nz, din, taps, dout);
    HIPCHK(c, hipGetLastError());
    return PBRT_OK;
}
static int log_enqueue(pbrt_ctx *c, uint32_t n, const float *din, float dr, float *dout) {
    float *mx = (float *)c->buf("img_max", ENV_MAX_BLOCKS * 4);
    if (!mx) return PBRT_E_NOMEM;
    ImgTimer tm(c, IMG_LOG);
    const uint32_t nb = std::max(1u, std::min<uint32_t>(div_up(n, 1024), ENV_MAX_BLOCKS));
    hipLaunchKernelGGL(k_env_max, dim3(nb), dim3(256), 0, c->stream, n, din, mx);
    hipLaunchKernelGGL(k_log_compress, dim3(div_up(n, 256)), dim3(256), 0, c->stream, n, din, mx, nb, dr, dout);
    HIPCHK(c, hipGetLastError());
    return PBRT_OK;
}
static int pulse_check(pbrt_ctx *ctx, uint32_t n_traces, uint32_t time_samples, float fs, float frequency, float sigma, uint32_t *K) {
    NEED(ctx, fs > 0.0f && frequency > 0.0f && sigma > 0.0f);
    NEED(ctx, (uint64_t)n_traces * time_samples < 0xffffffffull && n_traces <= 65535u);
    *K = (uint32_t)std::ceil(2.5 * (double)sigma * (double)fs);
    NEED(ctx, *K <= PULSE_MAX_K);
    return PBRT_OK;
}
static int pulse_enqueue(pbrt_ctx *c, uint32_t n_traces, uint32_t T, uint32_t K, float fs, float fc, float sigma, const float *din,
                         float *dout) {
    ImgTimer tm(c, IMG_PULSE);
    const size_t lds = (size_t)(2 * K + 1 + 256 + 2 * K) * 4;
    hipLaunchKernelGGL(k_apply_pulse, dim3(div_up(T, 256), n_traces), dim3(256), lds, c->stream, T, K, fs, fc, sigma, din, dout);
    HIPCHK(c, hipGetLastError());
    return PBRT_OK;
}

extern "C" {

int pbrt_das_beamform_dev(pbrt_ctx *ctx, const pbrt_das_params *p, const void *d_data, const void *d_tx_delays, const void *d_elem_x,
                          const void *d_x, const void *d_z, void *d_out) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, p && d_data && d_tx_delays && d_elem_x && d_x && d_z && d_out);
    int rc = das_check(ctx, p);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return das_enqueue(ctx, p, (const float *)d_data, (const float *)d_tx_delays, (const float *)d_elem_x, (const float *)d_x,
                       (const float *)d_z, (float *)d_out);
}

int pbrt_das_first_arrival_dev(pbrt_ctx *ctx, const pbrt_das_params *p, const void *d_tx_delays, const void *d_elem_x, const void *d_x,
                               const void *d_z, void *d_table) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, p && d_tx_delays && d_elem_x && d_x && d_z && d_table);
    int rc = das_check(ctx, p);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_das_first_arrival, dim3(div_up((uint64_t)p->nx * p->nz, 256)), dim3(256), 0, ctx->stream, *p, (const float *)d_tx_delays,
                       (const float *)d_elem_x, (const float *)d_x, (const float *)d_z, (double *)d_table);
    HIPCHK(ctx, hipGetLastError());
    return PBRT_OK;
}

int pbrt_das_beamform_table_dev(pbrt_ctx *ctx, const pbrt_das_params *p, const void *d_data, const void *d_table, const void *d_elem_x,
                                const void *d_x, const void *d_z, void *d_out) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, p && d_data && d_table && d_elem_x && d_x && d_z && d_out);
    int rc = das_check(ctx, p);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return das_enqueue(ctx, p, (const float *)d_data, nullptr, (const float *)d_elem_x, (const float *)d_x, (const float *)d_z,
                       (float *)d_out, (const double *)d_table);
}

int pbrt_das_beamform(pbrt_ctx *ctx, const pbrt_das_params *p, const float *data, const float *tx_delays,
                      const float *elem_x, const float *x, const float *z, float *out) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, p && data && tx_delays && elem_x && x && z && out);
    int rc = das_check(ctx, p);
    if (rc) return rc;
    const size_t nd = (size_t)p->n_angles * p->n_elements * p->time_samples, ne = (size_t)p->n_angles * p->n_elements;
    const uint32_t n = p->nx * p->nz;
    LEAF_BEGIN(ctx, (nd + ne + p->n_elements + p->nx + p->nz + (size_t)n) * 4 + 256);
    (void)grid;
    (void)block;
    (void)st;
    float *dd = S.in(data, nd), *dt = S.in(tx_delays, ne), *de = S.in(elem_x, p->n_elements);
    float *dx = S.in(x, p->nx), *dz = S.in(z, p->nz);
    float *dout = S.out<float>(n);
    if ((rc = das_enqueue(c, p, dd, dt, de, dx, dz, dout)) != 0) return rc;
    S.back(out, dout, n);
    return S.finish();
}

int pbrt_envelope_dev(pbrt_ctx *ctx, uint32_t nx, uint32_t nz, const void *d_rf, void *d_env) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, d_rf && d_env && d_rf != d_env && nz <= ENV_MAX_N && (uint64_t)nx * nz < 0xffffffffull);
    if (nx == 0 || nz == 0) return PBRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return env_enqueue(ctx, nx, nz, (const float *)d_rf, (float *)d_env);
}

int pbrt_envelope(pbrt_ctx *ctx, uint32_t nx, uint32_t nz, const float *rf, float *env) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, rf && env && nz <= ENV_MAX_N);
    const uint32_t n = nx * nz;
    NEED(ctx, (uint64_t)nx * nz < 0xffffffffull);
    LEAF_BEGIN(ctx, (size_t)n * 8 + 64);
    (void)grid;
    (void)block;
    (void)st;
    float *din = S.in(rf, n), *dout = S.out<float>(n);
    int rc = env_enqueue(c, nx, nz, din, dout);
    if (rc) return rc;
    S.back(env, dout, n);
    return S.finish();
}

int pbrt_log_compress_dev(pbrt_ctx *ctx, uint32_t n, const void *d_env, float dynamic_range_db, void *d_out) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, d_env && d_out && dynamic_range_db > 0.0f);
    if (n == 0) return PBRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return log_enqueue(ctx, n, (const float *)d_env, dynamic_range_db, (float *)d_out);
}

int pbrt_log_compress(pbrt_ctx *ctx, uint32_t n, const float *env, float dynamic_range_db, float *out) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, env && out && dynamic_range_db > 0.0f);
    LEAF_BEGIN(ctx, (size_t)n * 8 + 64);
    (void)grid;
    (void)block;
    (void)st;
    float *din = S.in(env, n), *dout = S.out<float>(n);
    int rc = log_enqueue(c, n, din, dynamic_range_db, dout);
    if (rc) return rc;
    S.back(out, dout, n);
    return S.finish();
}

int pbrt_us_apply_pulse_dev(pbrt_ctx *ctx, uint32_t n_traces, uint32_t time_samples, float fs, float frequency, float sigma,
                            const void *d_in, void *d_out) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, d_in && d_out && d_in != d_out);
    uint32_t K = 0;
    int rc = pulse_check(ctx, n_traces, time_samples, fs, frequency, sigma, &K);
    if (rc) return rc;
    if (n_traces == 0 || time_samples == 0) return PBRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return pulse_enqueue(ctx, n_traces, time_samples, K, fs, frequency, sigma, (const float *)d_in, (float *)d_out);
}

int pbrt_us_apply_pulse(pbrt_ctx *ctx, uint32_t n_traces, uint32_t time_samples, float fs, float frequency, float sigma,
                        const float *in, float *out) {
    if (!ctx) return PBRT_E_INVALID;
    NEED(ctx, in && out && in != out);
    uint32_t K = 0;
    int rc = pulse_check(ctx, n_traces, time_samples, fs, frequency, sigma, &K);
    if (rc) return rc;
    const uint32_t n = n_traces * time_samples;
    LEAF_BEGIN(ctx, (size_t)n * 8 + 64);
    (void)grid;
    (void)block;
    (void)st;
    float *din = S.in(in, n), *dout = S.out<float>(n);
    if ((rc = pulse_enqueue(c, n_traces, time_samples, K, fs, frequency, sigma, din, dout)) != 0) return rc;
    S.back(out, dout, n);
    return S.finish();
}

// ---- device buffers and the stream (ABI 5): what a caller needs to keep the us_render loop in HBM without another GPU library ----
int pbrt_ctx_synchronize(pbrt_ctx *c) {
    if (!c) return PBRT_E_INVALID;
    NOT_RECORDING(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ctx_settle(c);  // (a queued acquisition: its guard words are looked at now)
}

int pbrt_ctx_set_profiling(pbrt_ctx *c, int on) {
    if (!c) return PBRT_E_INVALID;
    c->profiling = on != 0;
    c->img_mask = 0;
    return PBRT_OK;
}

int pbrt_get_image_stats(pbrt_ctx *c, pbrt_image_stats *out) {
    if (!c || !out) return PBRT_E_INVALID;
    std::memset(out, 0, sizeof *out);
    NOT_RECORDING(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double *slot[IMG_STEPS] = {&out->pulse_ms, &out->das_ms, &out->envelope_ms, &out->log_ms};
    for (int i = 0; i < IMG_STEPS; ++i) {
        if (!(c->img_mask & (1u << i))) continue;
        float t = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&t, c->img_ev[2 * i], c->img_ev[2 * i + 1]));
        *slot[i] = t;
    }
    out->das_model_bytes = c->img_das_bytes;
    out->measured = c->img_mask;
    return PBRT_OK;
}

int pbrt_dev_alloc(pbrt_ctx *c, uint64_t bytes, void **out) {
    if (!c || !out) return PBRT_E_INVALID;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)std::max<uint64_t>(bytes, 16));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return c->fail(PBRT_E_NOMEM, "pbrt_dev_alloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    }
    *out = p;
    return PBRT_OK;
}

int pbrt_dev_free(pbrt_ctx *c, void *p) {
    if (!c) return PBRT_E_INVALID;
    if (!p) return PBRT_OK;
    NOT_RECORDING(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // queued work may still read or write it
    (void)ctx_settle(c);
    ++c->ws_epoch;  // (a recording may hold this pointer)
    HIPCHK(c, hipFree(p));
    return PBRT_OK;
}

int pbrt_dev_upload(pbrt_ctx *c, void *dst_dev, const void *src_host, uint64_t bytes) {
    if (!c) return PBRT_E_INVALID;
    NEED(c, (dst_dev && src_host) || bytes == 0);
    if (!bytes) return PBRT_OK;
    NOT_RECORDING(c);
    HIPCHK(c, hipSetDevice(c->device));
    // in stream order behind the queued kernels; a pageable source is staged before the call returns, so the caller may reuse it
    HIPCHK(c, hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    return PBRT_OK;
}

int pbrt_dev_download(pbrt_ctx *c, void *dst_host, const void *src_dev, uint64_t bytes) {
    if (!c) return PBRT_E_INVALID;
    NEED(c, (dst_host && src_dev) || bytes == 0);
    NOT_RECORDING(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (bytes) HIPCHK(c, hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ctx_settle(c);  // an acquisition queued ahead of this copy that tripped its guard makes the copy's content invalid
}

// ---- the queued chain as one submission (include/pbrt_hip.h) ------------------------------------------------------------------
struct pbrt_graph {
    pbrt_ctx *ctx = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    pbrt_ctx::PendingAcq pend;  // what the recorded acquisition leaves for us_finish (inactive: the recording holds none)
    uint64_t epoch = 0;         // ctx->ws_epoch at the end of the recording
    uint64_t das_bytes = 0;
    const void *rows = nullptr; // the counter rows the recorded acquisition expects clean (it was recorded without their fill command)
    size_t rows_bytes = 0;
};

int pbrt_ctx_record_begin(pbrt_ctx *c) {
    if (!c) return PBRT_E_INVALID;
    if (int rc = ctx_settle(c)) return rc;  // (also: no recording inside a recording)
    HIPCHK(c, hipSetDevice(c->device));
    // relaxed: the recorded entry points query function attributes and free memory, calls a stricter mode refuses on any thread
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    c->recording = true;
    c->rec_failed = false;
    c->rec_msg.clear();
    return PBRT_OK;
}

int pbrt_ctx_record_end(pbrt_ctx *c, pbrt_graph **out) {
    if (!c || !out) return PBRT_E_INVALID;
    *out = nullptr;
    if (!c->recording) return c->fail(PBRT_E_INVALID, "pbrt_ctx_record_end without pbrt_ctx_record_begin");
    c->recording = false;
    pbrt_ctx::PendingAcq P = c->pend;
    c->pend.active = false;  // (nothing ran)
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (e != hipSuccess || !g) {
        (void)hipGetLastError();
        return c->fail(PBRT_E_DEVICE, "hipStreamEndCapture: %s (a call inside the recording failed or waited)", hipGetErrorString(e));
    }
    if (c->rec_failed) {  // (the capture itself is intact, but what it holds is not the chain the caller meant)
        (void)hipGraphDestroy(g);
        return c->fail(PBRT_E_INVALID, "a call inside the recording failed: %s", c->rec_msg.c_str());
    }
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(g);
        return c->fail(PBRT_E_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    pbrt_graph *G = new pbrt_graph;
    G->ctx = c;
    G->graph = g;
    G->exec = x;
    G->pend = P;
    G->pend.timed = false;
    G->epoch = c->ws_epoch;
    G->das_bytes = c->img_das_bytes;
    if (P.active) {
        G->rows = c->us_rows_clean;
        G->rows_bytes = c->us_rows_clean_bytes;
    }
    ++c->n_graphs;
    *out = G;
    return PBRT_OK;
}

int pbrt_graph_launch(pbrt_graph *G) {
    if (!G) return PBRT_E_INVALID;
    pbrt_ctx *c = G->ctx;
    if (int rc = ctx_settle(c)) return rc;
    if (G->epoch != c->ws_epoch)
        return c->fail(PBRT_E_INVALID, "the recording is stale: memory or tables it refers to were freed or replaced since it was made; record again");
    if (G->pend.active && (c->us_rows_clean != G->rows || c->us_rows_clean_bytes != G->rows_bytes))
        return c->fail(PBRT_E_INVALID, "the recording is stale: another acquisition (or one that failed) used the counters since it was made; record again");
    HIPCHK(c, hipSetDevice(c->device));
    ++c->call_seq;
    for (auto &kv : c->ws) kv.second.stamp = c->call_seq;  // (a trim between launches must not take what the replay uses)
    HIPCHK(c, hipGraphLaunch(G->exec, c->stream));
    c->pend = G->pend;
    c->img_das_bytes = G->das_bytes;
    return PBRT_OK;
}

int pbrt_graph_destroy(pbrt_graph *G) {
    if (!G) return PBRT_OK;
    pbrt_ctx *c = G->ctx;
    (void)hipSetDevice(c->device);
    if (!c->recording) (void)hipStreamSynchronize(c->stream);  // a replay may still run
    if (G->exec) (void)hipGraphExecDestroy(G->exec);
    if (G->graph) (void)hipGraphDestroy(G->graph);
    if (c->n_graphs) --c->n_graphs;
    delete G;
    return PBRT_OK;
}

}  // extern "C"
