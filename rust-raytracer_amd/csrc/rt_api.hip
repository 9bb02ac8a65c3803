// C ABI of the MI355X render hot path (declarations and reference citations: include/rtamd.h).
// Host side only: context, scene upload (reference layout -> device layout), launch sequencing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "rt_kernels.h"
#include "rtamd.h"

using namespace rtamd;

struct rtamd_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    LaunchCfg cfg;
    bool profile = false;
    bool static_share_set = false;   // RTAMD_OPT_STATIC_SHARE was given: the upload does not pick it
    int deepen = 1;          // object-tree walks in passes of growing reach (rt_sched.h): 0 never, 1 in the render entry points, 2 also in the batch entry points
    int ordered = 1;         // walks that visit the nearer child first (rt_sched.h): same three values
    uint32_t ordered_ok = 0; // ORD_MESH / ORD_TREE: the uploaded scene's trees allow them
    uint32_t ordered_cap = 15; // stack entries an ordered walk may use (RTAMD_OPT_ORDERED_STACK)
    int frame_kernel = 2;    // RTAMD_OPT_FRAME_KERNEL: 0 never, 1 wherever it can run, 2 where it is the faster form
    int frame_step_min = 64; // RTAMD_OPT_FRAME_STEP_MIN
    int frame_rebalance = 0; // RTAMD_OPT_FRAME_REBALANCE
    bool df_rounds = true;   // scatter_rays != 1: rounds of scheduler-kernel launches (false: one lane per pixel, nested loops)
    size_t max_paths = (size_t)1 << 25;   // pixel slots per wavefront chunk: 33.5 M slots (an 8K frame in one chunk), about 470 B each at 3 light samples

    // scene
    bool have_scene = false;
    DevScene S{};
    std::vector<void*> scene_allocs;
    uint32_t total_samples_lights = 0;   // not used on the host; kept for introspection
    std::vector<rtamd_light> lights;

    // wavefront workspaces: one per pixel group in flight, each with its own stream, so that the
    // short late-bounce launches of one group overlap the long launches of another
    struct Workspace {
        WaveBufs W{};
        std::vector<void*> allocs;
        size_t cap = 0;
        uint32_t samples = 0;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
    };
    std::vector<Workspace> ws;
    int n_streams = 1;

    // ray batch workspace of the host-pointer entry points
    void* batch_buf = nullptr;
    size_t batch_bytes = 0;
    // Queue heads of the batch trace kernels: a ring of HEAD_RING words, one per call in turn, so that calls in flight
    // on different streams of the caller do not share a head.
    uint32_t* work_head = nullptr;
    uint32_t head_next = 0;

    // kernel timing (RTAMD_OPT_PROFILE)
    double closest_ms = 0, any_ms = 0, shade_ms = 0;
    uint64_t closest_launches = 0, any_launches = 0;
    struct Timed { hipEvent_t a, b; int kind; };
    std::vector<Timed> timed;
    std::vector<hipEvent_t> event_pool;
};

// Queue heads of rtamd_trace_closest_static, the entry point that allocates nothing: a ring of HEAD_RING words in the
// library's own image.  Every GPU the library runs on holds its own copy at its own address; the name below is only a
// host-side handle for hipGetSymbolAddress, host code uses the per-device pointers in g_scratch_ptr.
static constexpr uint32_t HEAD_RING = 64;
static __device__ __align__(256) uint32_t scratch[HEAD_RING];

namespace {

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
uint32_t* g_scratch_ptr[MAX_DEVICES] = {};   // device address of `scratch` on each GPU, set by rtamd_create
uint32_t g_scratch_next[MAX_DEVICES] = {};   // next ring slot per GPU

int fail(rtamd_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg;
    return code;
}
#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, RTAMD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

void free_list(std::vector<void*>& v) {
    for (void* p : v) (void)hipFree(p);
    v.clear();
}

template <class T>
int upload_vec(rtamd_context* ctx, const std::vector<T>& host, const T** dev) {
    size_t bytes = host.size() * sizeof(T);
    void* p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, bytes ? bytes : 16));
    ctx->scene_allocs.push_back(p);
    if (bytes) HIP_TRY(ctx, hipMemcpy(p, host.data(), bytes, hipMemcpyHostToDevice));
    *dev = (const T*)p;
    return RTAMD_OK;
}

// Reference layout (AABB, skip, first) + sentinel  ->  device nodes (see rt_device.h).
// *plain is cleared when a box is not finite or has min > max on some axis (what the kernels' NaN-free box test presumes).
int convert_tree(rtamd_context* ctx, const rtamd_bvh_node* nodes, uint32_t n_nodes, uint32_t n_prims, const char* what,
                 std::vector<float4>& out, bool* plain = nullptr) {
    if (nodes == nullptr || n_nodes < 2) return fail(ctx, RTAMD_E_INVALID, std::string(what) + ": a tree needs at least one node and the sentinel");
    uint32_t n = n_nodes - 1;
    if (nodes[n].first != n_prims) return fail(ctx, RTAMD_E_INVALID, std::string(what) + ": sentinel.first must equal the primitive count");
    for (uint32_t i = 0; i < n; i++) {
        const rtamd_bvh_node& nd = nodes[i];
        uint32_t first = nd.first, end = nodes[i + 1].first;
        if (nd.skip <= i || nd.skip > n) return fail(ctx, RTAMD_E_INVALID, std::string(what) + ": skip link must point forward and inside the tree");
        if (first > end || end > n_prims) return fail(ctx, RTAMD_E_INVALID, std::string(what) + ": primitive range out of order");
        bool leaf = nd.skip == i + 1;
        if (!leaf && first != end) return fail(ctx, RTAMD_E_INVALID, std::string(what) + ": inner node with a primitive range");
        uint32_t A = leaf ? (LEAF_BIT | first) : nd.skip;
        uint32_t B = leaf ? end : 0u;
        if (leaf && (first & LEAF_BIT)) return fail(ctx, RTAMD_E_INVALID, std::string(what) + ": too many primitives");
        if (plain)
            for (int k = 0; k < 3; k++)
                if (!(std::isfinite(nd.min[k]) && std::isfinite(nd.max[k]) && nd.min[k] <= nd.max[k])) *plain = false;
        float fa, fb;
        std::memcpy(&fa, &A, 4);
        std::memcpy(&fb, &B, 4);
        out.push_back(make_float4(nd.min[0], nd.min[1], nd.min[2], nd.max[0]));
        out.push_back(make_float4(nd.max[1], nd.max[2], fa, fb));
    }
    return RTAMD_OK;
}

// The padded box of an object in the space of the scene's rays (see DevObject): the primitive's own bounds carried out
// through the wrappers, innermost first, in double precision, then padded by 1e-4 of its size (the float rounding of
// the reference's own transforms is orders of magnitude below that).  slack = 0.011 x the chain's scale: the visit test
// of the trees lets t run 0.01 beyond the box in the object's t (bvh.rs:231), and t grows by `scale` on the way out
// (transform.rs:104-107).
void object_box(const rtamd_scene_desc* sc, const rtamd_object& ob, const std::vector<float4>& nodes, const std::vector<DevMesh>& meshes, DevObject& d,
                bool zero_translate = false /* the box relative to the chain's offset: every Translate counts as zero */) {
    const float inf = std::numeric_limits<float>::infinity();
    for (int k = 0; k < 3; k++) { d.bmin[k] = -inf; d.bmax[k] = inf; }
    d.slack = inf;
    double lo[3] = {-1.0, -1.0, -1.0}, hi[3] = {1.0, 1.0, 1.0};   // unit sphere, unit cuboid
    if (ob.geom_kind == RTAMD_GEOM_MESH) {
        const float4 n0 = nodes[2 * (size_t)meshes[ob.geom_index].node_base], n1 = nodes[2 * (size_t)meshes[ob.geom_index].node_base + 1];
        lo[0] = n0.x; lo[1] = n0.y; lo[2] = n0.z; hi[0] = n0.w; hi[1] = n1.x; hi[2] = n1.y;
    } else if (ob.geom_kind == RTAMD_GEOM_CUBOID) {
        const rtamd_cuboid& c = sc->cuboids[ob.geom_index];
        for (int k = 0; k < 3; k++) { lo[k] = (double)c.center[k] - std::fabs((double)c.extent[k]); hi[k] = (double)c.center[k] + std::fabs((double)c.extent[k]); }
    } else if (ob.geom_kind == RTAMD_GEOM_TRIANGLE) {
        const rtamd_triangle& t = sc->triangles[ob.geom_index];
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min({(double)t.a[k], (double)t.b[k], (double)t.c[k]});
            hi[k] = std::max({(double)t.a[k], (double)t.b[k], (double)t.c[k]});
        }
    }
    double scale = 1.0;
    for (uint32_t q = ob.op_count; q-- > 0;) {
        const rtamd_xform_op& op = sc->ops[ob.op_first + q];
        if (op.kind == RTAMD_OP_TRANSLATE) {
            if (!zero_translate)
                for (int k = 0; k < 3; k++) { lo[k] += op.v[k]; hi[k] += op.v[k]; }
        } else if (op.kind == RTAMD_OP_SCALE) {
            if (!(op.v[0] > 0.0f)) return;   // mirrored or degenerate: no box
            scale *= op.v[0];
            for (int k = 0; k < 3; k++) { lo[k] *= op.v[0]; hi[k] *= op.v[0]; }
        } else if (op.kind == RTAMD_OP_ROTATE) {
            double nlo[3] = {1e300, 1e300, 1e300}, nhi[3] = {-1e300, -1e300, -1e300};
            for (int c = 0; c < 8; c++) {
                const double p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
                for (int i = 0; i < 3; i++) {
                    const double v = (double)op.v[3 * i] * p[0] + (double)op.v[3 * i + 1] * p[1] + (double)op.v[3 * i + 2] * p[2];
                    nlo[i] = std::min(nlo[i], v); nhi[i] = std::max(nhi[i], v);
                }
            }
            for (int k = 0; k < 3; k++) { lo[k] = nlo[k]; hi[k] = nhi[k]; }
        }   // BOUNDS: the object's own bounds are what counts
    }
    double size = 0.0;
    for (int k = 0; k < 3; k++) {
        if (!std::isfinite(lo[k]) || !std::isfinite(hi[k]) || lo[k] > hi[k]) return;
        size = std::max({size, hi[k] - lo[k], std::fabs(lo[k]), std::fabs(hi[k])});
    }
    const double pad = 1e-4 * size + 1e-6;
    for (int k = 0; k < 3; k++) {
        d.bmin[k] = std::nextafter((float)(lo[k] - pad), -inf);
        d.bmax[k] = std::nextafter((float)(hi[k] + pad), inf);
    }
    d.slack = (float)(0.011 * scale + 1e-6);
    // worth testing where entering is costly and missing is likely: a mesh under a rotation
    bool rotated = false;
    for (uint32_t q = 0; q < ob.op_count; q++) rotated = rotated || sc->ops[ob.op_first + q].kind == RTAMD_OP_ROTATE;
    if (ob.geom_kind == RTAMD_GEOM_MESH && rotated) d.xf_kind |= XF_BOX;
}

int ensure_batch(rtamd_context* ctx, size_t bytes) {
    if (ctx->batch_bytes >= bytes) return RTAMD_OK;
    if (ctx->batch_buf) (void)hipFree(ctx->batch_buf);
    ctx->batch_buf = nullptr; ctx->batch_bytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->batch_buf, bytes));
    ctx->batch_bytes = bytes;
    return RTAMD_OK;
}

uint32_t light_samples(const rtamd_light& l, int shadow_rays) {
    bool delta = l.kind == RTAMD_LIGHT_AMBIENT || (l.kind == RTAMD_LIGHT_POINT && l.param == 0.0f) ||
                 (l.kind == RTAMD_LIGHT_DIRECTION && l.param > 0.0f);
    return delta ? 1u : (uint32_t)(shadow_rays > 0 ? shadow_rays : 0);
}

template <class T>
int wave_alloc(rtamd_context* ctx, rtamd_context::Workspace& ws, T** p, size_t count) {
    void* q = nullptr;
    HIP_TRY(ctx, hipMalloc(&q, count * sizeof(T) + 256));
    ws.allocs.push_back(q);
    *p = (T*)q;
    return RTAMD_OK;
}

int ensure_wave(rtamd_context* ctx, rtamd_context::Workspace& ws, size_t cap, uint32_t samples) {
    if (!ws.stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ws.stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ws.done, hipEventDisableTiming));
    }
    if (ws.cap >= cap && ws.samples >= samples) return RTAMD_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ws.stream));
    free_list(ws.allocs);
    ws.cap = 0; ws.samples = 0;
    WaveBufs& W = ws.W;
    int rc;
    const size_t S = samples ? samples : 1;
    if (cap * S > 0xFFFFFFF0ull) return fail(ctx, RTAMD_E_INVALID, "chunk size x light samples exceeds 2^32; lower RTAMD_OPT_MAX_PATHS");
    W.cap = (uint32_t)cap;
    W.any_cap = (uint32_t)(cap * S);
    if ((rc = wave_alloc(ctx, ws, &W.rng_lo, cap))) return rc;
    if ((rc = wave_alloc(ctx, ws, &W.rng_hi, cap))) return rc;
    if ((rc = wave_alloc(ctx, ws, &W.sum, 3 * cap))) return rc;
    for (int k = 0; k < 2; k++) {
        if ((rc = wave_alloc(ctx, ws, &W.Lb[k], 3 * cap))) return rc;
        if ((rc = wave_alloc(ctx, ws, &W.Lpb[k], 3 * cap))) return rc;
    }
    W.L = W.Lb[0];
    W.Lp = W.Lpb[0];
    if ((rc = wave_alloc(ctx, ws, &W.beta, 3 * cap))) return rc;
    for (int k = 0; k < 2; k++) {
        if ((rc = wave_alloc(ctx, ws, &W.q_slot[k], cap))) return rc;
        if ((rc = wave_alloc(ctx, ws, &W.q_ray[k], 6 * cap))) return rc;
    }
    if ((rc = wave_alloc(ctx, ws, &W.h_rec, 2 * cap))) return rc;
    if ((rc = wave_alloc(ctx, ws, &W.p_slot, cap))) return rc;
    if ((rc = wave_alloc(ctx, ws, &W.p_ray, 6 * cap))) return rc;
    if ((rc = wave_alloc(ctx, ws, &W.p_rec, 2 * cap))) return rc;
    if ((rc = wave_alloc(ctx, ws, &W.p_pix, cap))) return rc;
    for (int k = 0; k < 4; k++) {
        ShadeSet& T = W.set[k];
        if ((rc = wave_alloc(ctx, ws, &T.lit_slot, cap))) return rc;
        if ((rc = wave_alloc(ctx, ws, &T.lit_f, 6 * cap))) return rc;
        if ((rc = wave_alloc(ctx, ws, &T.smp, 4 * S * cap))) return rc;
        if ((rc = wave_alloc(ctx, ws, &T.a_ray, 7 * S * cap))) return rc;
        if ((rc = wave_alloc(ctx, ws, &T.a_vidx, S * cap))) return rc;
        if ((rc = wave_alloc(ctx, ws, &T.vis, S * cap))) return rc;
    }
    if ((rc = wave_alloc(ctx, ws, &W.counts, (size_t)CounterSlots::WORDS))) return rc;
    if ((rc = wave_alloc(ctx, ws, &W.stats, (size_t)16))) return rc;
    ws.cap = cap; ws.samples = samples;
    return RTAMD_OK;
}

hipEvent_t get_event(rtamd_context* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void timed_begin(rtamd_context* ctx, int kind, hipStream_t st) {
    if (!ctx->profile) return;
    rtamd_context::Timed t{get_event(ctx), get_event(ctx), kind};
    (void)hipEventRecord(t.a, st);
    ctx->timed.push_back(t);
}
void timed_end(rtamd_context* ctx, hipStream_t st) {
    if (!ctx->profile) return;
    (void)hipEventRecord(ctx->timed.back().b, st);
}
// Call after the stream is synchronised.
void timed_collect(rtamd_context* ctx, rtamd_stats* stats) {
    double c = 0, a = 0, s = 0;
    for (auto& t : ctx->timed) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, t.a, t.b);
        if (t.kind == 0) { c += ms; ctx->closest_launches++; }
        else if (t.kind == 1) { a += ms; ctx->any_launches++; }
        else s += ms;
        ctx->event_pool.push_back(t.a);
        ctx->event_pool.push_back(t.b);
    }
    ctx->timed.clear();
    ctx->closest_ms += c; ctx->any_ms += a; ctx->shade_ms += s;
    if (stats) { stats->trace_closest_ms = c; stats->trace_any_ms = a; stats->shade_ms = s; }
}

// squares: the wavefront pipelines take their pixels in squares of 8 x 8 where the tile allows it (headline frame 57.1 -> 56.2 ms:
// the shade launches read closer together); the frame kernel takes them in rows (a wave that owns five squares of the model
// and another that owns five of the sky are further apart than waves that own five lines through both: 60.5 -> 64.9 ms with squares).
#ifndef RTAMD_SLOT_SQUARES
#define RTAMD_SLOT_SQUARES 1
#endif
RegionDev make_region(const rtamd_region& r, bool squares = true) {
    RegionDev R;
    R.squares = (RTAMD_SLOT_SQUARES != 0 && squares && ((r.tile_w | r.tile_h) & 7u) == 0u) ? 1u : 0u;
    R.x0 = r.x0; R.y0 = r.y0; R.w = r.w; R.h = r.h;
    R.tile_w = r.tile_w; R.tile_h = r.tile_h; R.tile_first = r.tile_first; R.tile_stride = r.tile_stride;
    R.tiles_x = (r.w + r.tile_w - 1) / r.tile_w;
    uint32_t tiles_y = (r.h + r.tile_h - 1) / r.tile_h;
    R.n_tiles = R.tiles_x * tiles_y;
    return R;
}
bool region_ok(const rtamd_region* r) {
    return r && r->w > 0 && r->h > 0 && r->tile_w > 0 && r->tile_h > 0 && r->tile_stride > 0 && r->tile_first < r->tile_stride;
}
size_t owned_tiles(const rtamd_region& r) {
    size_t tiles_x = (r.w + r.tile_w - 1) / r.tile_w, tiles_y = (r.h + r.tile_h - 1) / r.tile_h;
    size_t n = tiles_x * tiles_y;
    if (r.tile_first >= n) return 0;
    return (n - r.tile_first + r.tile_stride - 1) / r.tile_stride;
}
// valid (inside-rectangle) pixels among the owned tiles
uint64_t owned_pixels(const rtamd_region& r) {
    uint64_t tiles_x = (r.w + r.tile_w - 1) / r.tile_w, tiles_y = (r.h + r.tile_h - 1) / r.tile_h;
    uint64_t n = tiles_x * tiles_y, px = 0;
    for (uint64_t t = r.tile_first; t < n; t += r.tile_stride) {
        uint64_t ty = t / tiles_x, tx = t % tiles_x;
        uint64_t w = std::min<uint64_t>(r.tile_w, r.w - tx * r.tile_w), h = std::min<uint64_t>(r.tile_h, r.h - ty * r.tile_h);
        px += w * h;
    }
    return px;
}

}  // namespace

extern "C" {

int rtamd_create(int device_id, rtamd_context** out) {
    if (!out) return RTAMD_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return RTAMD_E_DEVICE;
    rtamd_context* ctx = new rtamd_context();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return RTAMD_E_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) ctx->cfg.num_cus = prop.multiProcessorCount;
    if (const char* e = std::getenv("RTAMD_THIN_TAIL")) ctx->cfg.thin_tail = std::atoi(e) != 0 ? 1 : 0;   // (A/B of the scheduler's deal of mid-size queues)
    if (hipMalloc((void**)&ctx->work_head, HEAD_RING * sizeof(uint32_t)) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return RTAMD_E_DEVICE;
    }
    {   // this GPU's address of the static scratch array (once per device, with the device current)
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (device_id < MAX_DEVICES && g_scratch_ptr[device_id] == nullptr) {
            void* p = nullptr;
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || p == nullptr) {
                (void)hipFree(ctx->work_head);
                (void)hipStreamDestroy(ctx->stream);
                delete ctx;
                return RTAMD_E_DEVICE;
            }
            g_scratch_ptr[device_id] = (uint32_t*)p;
        }
    }
    *out = ctx;
    return RTAMD_OK;
}

void rtamd_destroy(rtamd_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_list(ctx->scene_allocs);
    for (auto& w : ctx->ws) {
        if (w.stream) (void)hipStreamSynchronize(w.stream);
        free_list(w.allocs);
        if (w.done) (void)hipEventDestroy(w.done);
        if (w.stream) (void)hipStreamDestroy(w.stream);
    }
    if (ctx->batch_buf) (void)hipFree(ctx->batch_buf);
    if (ctx->work_head) (void)hipFree(ctx->work_head);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* rtamd_last_error(const rtamd_context* ctx) { return ctx ? ctx->error.c_str() : "no context"; }

int rtamd_set_option(rtamd_context* ctx, int option, int64_t value) {
    if (!ctx) return RTAMD_E_INVALID;
    switch (option) {
        case RTAMD_OPT_PROFILE: ctx->profile = value != 0; return RTAMD_OK;
        case RTAMD_OPT_MAX_PATHS:
            if (value < 64) return fail(ctx, RTAMD_E_INVALID, "max_paths must be >= 64");
            ctx->max_paths = (size_t)value;
            return RTAMD_OK;
        case RTAMD_OPT_LDS_SCENE: ctx->cfg.use_lds = value != 0; return RTAMD_OK;
        case RTAMD_OPT_DEPTH_FIRST_ROUNDS: ctx->df_rounds = value != 0; return RTAMD_OK;
        case RTAMD_OPT_LEAN_KERNELS: ctx->cfg.lean = value != 0 ? 1 : 0; return RTAMD_OK;
        case RTAMD_OPT_TREE_DEEPENING:
            if (value < 0 || value > 2) return fail(ctx, RTAMD_E_INVALID, "tree deepening mode must be 0, 1 or 2");
            ctx->deepen = (int)value; ctx->S.deepen = ctx->deepen >= 1 ? 1u : 0u;
            return RTAMD_OK;
        case RTAMD_OPT_ORDERED_WALK:
            if (value < 0 || value > 2) return fail(ctx, RTAMD_E_INVALID, "ordered walk mode must be 0, 1 or 2");
            ctx->ordered = (int)value; ctx->S.ordered = ctx->ordered >= 1 ? ctx->ordered_ok : 0u;
            return RTAMD_OK;
        case RTAMD_OPT_ORDERED_STACK:
            if (value < 0 || value > 15) return fail(ctx, RTAMD_E_INVALID, "ordered stack entries must be in [0, 15]");
            ctx->ordered_cap = (uint32_t)value; ctx->S.ordered_cap = ctx->ordered_cap;
            return RTAMD_OK;
        case RTAMD_OPT_FRAME_KERNEL:
            if (value < 0 || value > 2) return fail(ctx, RTAMD_E_INVALID, "frame kernel mode must be 0, 1 or 2");
            ctx->frame_kernel = (int)value;
            return RTAMD_OK;
        case RTAMD_OPT_FRAME_STEP_MIN:
            if (value < 1 || value > 64) return fail(ctx, RTAMD_E_INVALID, "frame step threshold must be in [1, 64]");
            ctx->frame_step_min = (int)value;
            return RTAMD_OK;
        case RTAMD_OPT_FRAME_REBALANCE:
            if (value < 0 || value > 6) return fail(ctx, RTAMD_E_INVALID, "frame rebalance count must be in [0, 6]");
            ctx->frame_rebalance = (int)value;
            return RTAMD_OK;
        case RTAMD_OPT_STRICT_ORDER:
            ctx->deepen = value != 0 ? 0 : 1; ctx->S.deepen = ctx->deepen >= 1 ? 1u : 0u;
            ctx->ordered = value != 0 ? 0 : 1; ctx->S.ordered = ctx->ordered >= 1 ? ctx->ordered_ok : 0u;
            return RTAMD_OK;
        case RTAMD_OPT_SCHEDULER: ctx->cfg.scheduler = value != 0; return RTAMD_OK;
        case RTAMD_OPT_CHUNK:
            if (value < 64 || value > 4096 || (value & (value - 1))) return fail(ctx, RTAMD_E_INVALID, "chunk must be a power of two in [64, 4096]");
            ctx->cfg.chunk = (int)value;
            return RTAMD_OK;
        case RTAMD_OPT_BLOCKS_PER_CU:
            if (value < 1 || value > 8) return fail(ctx, RTAMD_E_INVALID, "blocks per CU must be in [1, 8]");
            ctx->cfg.sched_blocks_per_cu = (int)value;
            return RTAMD_OK;
        case RTAMD_OPT_STATIC_SHARE:
            if (value < 0 || value > 8) return fail(ctx, RTAMD_E_INVALID, "static share must be in [0, 8] eighths");
            ctx->cfg.static_share = (int)value;
            ctx->static_share_set = true;
            return RTAMD_OK;
        case RTAMD_OPT_MIN_BATCH:
            if (value < 1 || value > 64) return fail(ctx, RTAMD_E_INVALID, "min batch must be in [1, 64]");
            ctx->cfg.min_batch = (int)value;
            return RTAMD_OK;
        case RTAMD_OPT_STREAMS:
            if (value < 1 || value > 8) return fail(ctx, RTAMD_E_INVALID, "streams must be in [1, 8]");
            ctx->n_streams = (int)value;
            return RTAMD_OK;
        default: return fail(ctx, RTAMD_E_INVALID, "unknown option");
    }
}

int rtamd_kernel_times(rtamd_context* ctx, double* closest_ms_total, uint64_t* closest_launches, double* any_ms_total,
                       uint64_t* any_launches, int reset) {
    if (!ctx) return RTAMD_E_INVALID;
    if (closest_ms_total) *closest_ms_total = ctx->closest_ms;
    if (closest_launches) *closest_launches = ctx->closest_launches;
    if (any_ms_total) *any_ms_total = ctx->any_ms;
    if (any_launches) *any_launches = ctx->any_launches;
    if (reset) { ctx->closest_ms = ctx->any_ms = ctx->shade_ms = 0; ctx->closest_launches = ctx->any_launches = 0; }
    return RTAMD_OK;
}

int rtamd_upload_scene(rtamd_context* ctx, const rtamd_scene_desc* sc) {
    if (!ctx || !sc) return RTAMD_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_list(ctx->scene_allocs);
    ctx->have_scene = false;
    if (sc->n_root == 0 || !sc->root) return fail(ctx, RTAMD_E_INVALID, "scene root is empty");
    if (sc->n_objects == 0 || !sc->objects) return fail(ctx, RTAMD_E_INVALID, "scene has no objects");
    if (sc->n_materials == 0 || !sc->materials) return fail(ctx, RTAMD_E_INVALID, "scene has no materials");
    if ((sc->n_meshes && !sc->meshes) || (sc->n_cuboids && !sc->cuboids) || (sc->n_triangles && !sc->triangles) || (sc->n_ops && !sc->ops) ||
        (sc->n_trees && !sc->trees) || (sc->n_lights && !sc->lights) || (sc->n_images && !sc->images))
        return fail(ctx, RTAMD_E_INVALID, "a scene array is null although its count is not zero");
    for (uint32_t k = 0; k < sc->n_ops; k++) {
        const rtamd_xform_op& op = sc->ops[k];
        // Scale::new asserts scale > 0 (transform.rs:87); the pair must be the reference's (s, 1.0 / s)
        if (op.kind == RTAMD_OP_SCALE && !(op.v[0] > 0.0f && std::isfinite(op.v[0]) && op.v[1] == 1.0f / op.v[0]))
            return fail(ctx, RTAMD_E_INVALID, "SCALE op: v[0] must be a positive finite scale and v[1] its reciprocal 1.0f / v[0]");
    }

    std::vector<float4> nodes, tris, normals;
    std::vector<DevMesh> meshes;
    std::vector<DevTree> trees;
    bool plain_boxes = true, plain_tree_boxes = true;   // every box of the mesh trees / the object trees finite with min <= max
    int rc;
    {
        size_t nn = 0, nt = 0, nv = 0;
        for (uint32_t m = 0; m < sc->n_meshes; m++) { nn += sc->meshes[m].n_nodes; nt += sc->meshes[m].n_tris; nv += sc->meshes[m].n_verts; }
        for (uint32_t t = 0; t < sc->n_trees; t++) nn += sc->trees[t].n_nodes;
        nodes.reserve(2 * nn); tris.reserve(3 * nt); normals.reserve(nv); meshes.reserve(sc->n_meshes);
    }
    for (uint32_t m = 0; m < sc->n_meshes; m++) {
        const rtamd_mesh& me = sc->meshes[m];
        if (!me.tri_indices || !me.vertices || !me.normals) return fail(ctx, RTAMD_E_INVALID, "mesh with null arrays");
        DevMesh dm;
        dm.node_base = (uint32_t)(nodes.size() / 2);
        dm.n_nodes = me.n_nodes ? me.n_nodes - 1 : 0;
        dm.tri_base = (uint32_t)(tris.size() / 3);
        dm.normal_base = (uint32_t)normals.size();
        if ((rc = convert_tree(ctx, me.nodes, me.n_nodes, me.n_tris, "mesh tree", nodes, &plain_boxes))) return rc;
        for (uint32_t k = 0; k < me.n_tris; k++) {
            uint32_t i0 = me.tri_indices[3 * k], i1 = me.tri_indices[3 * k + 1], i2 = me.tri_indices[3 * k + 2];
            if (i0 >= me.n_verts || i1 >= me.n_verts || i2 >= me.n_verts) return fail(ctx, RTAMD_E_INVALID, "triangle vertex index out of bounds");
            const float *a = me.vertices + 3 * (size_t)i0, *b = me.vertices + 3 * (size_t)i1, *c = me.vertices + 3 * (size_t)i2;
            // ab = b - a, ac = c - a: src/geom/triangle.rs:21-22, hoisted out of the per-ray work
            float ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
            float ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            float f0, f1, f2;
            std::memcpy(&f0, &i0, 4); std::memcpy(&f1, &i1, 4); std::memcpy(&f2, &i2, 4);
            tris.push_back(make_float4(a[0], a[1], a[2], ab[0]));
            tris.push_back(make_float4(ab[1], ab[2], ac[0], ac[1]));
            tris.push_back(make_float4(ac[2], f0, f1, f2));
        }
        for (uint32_t v = 0; v < me.n_verts; v++)
            normals.push_back(make_float4(me.normals[3 * (size_t)v], me.normals[3 * (size_t)v + 1], me.normals[3 * (size_t)v + 2], 0.0f));
        meshes.push_back(dm);
    }
    for (uint32_t t = 0; t < sc->n_trees; t++) {
        const rtamd_object_tree& tr = sc->trees[t];
        if ((uint64_t)tr.first_object + tr.n_objects > sc->n_objects) return fail(ctx, RTAMD_E_INVALID, "object tree range outside objects[]");
        DevTree dt;
        dt.node_base = (uint32_t)(nodes.size() / 2);
        dt.n_nodes = tr.n_nodes ? tr.n_nodes - 1 : 0;
        dt.first_object = tr.first_object;
        // walks in passes of growing reach (rt_sched.h): 1 a tree of mesh objects, 2 a tree of cheap objects (limits only its
        // long rays), 0 a small tree
        uint32_t n_mesh = 0;
        for (uint32_t k = 0; k < tr.n_objects; k++) n_mesh += sc->objects[tr.first_object + k].geom_kind == RTAMD_GEOM_MESH ? 1u : 0u;
        dt.deepen = tr.n_nodes >= 257u ? (2u * (uint64_t)n_mesh >= tr.n_objects ? 1u : 2u) : 0u;
        // A walk that visits the nearer child first reports what the reference's does as long as a primitive's answer does not
        // depend on the t_max it is asked with beyond `t <= t_max`; a cuboid that holds the ray's origin answers its far side even
        // beyond t_max (cuboid.rs:22-28), so a tree with cuboids keeps the reference's order.
        bool orderable = tr.n_nodes >= 65u && tr.n_nodes <= (1u << 17);   // (the compact stack entries hold 17 bits of node index)
        for (uint32_t k = 0; k < tr.n_objects && orderable; k++) {
            const uint32_t g = sc->objects[tr.first_object + k].geom_kind;
            orderable = g == RTAMD_GEOM_MESH || g == RTAMD_GEOM_UNIT_SPHERE || g == RTAMD_GEOM_TRIANGLE;
        }
        if (orderable) dt.deepen |= TREE_ORDERED;
        if ((rc = convert_tree(ctx, tr.nodes, tr.n_nodes, tr.n_objects, "object tree", nodes, &plain_tree_boxes))) return rc;
        trees.push_back(dt);
    }
    // Two-child records for the ordered walks (see DevScene::wide): inner node i's record = its children's 32-byte records,
    // node i + 1 and that node's skip link (or i + 2 behind a leaf).  A tree some inner node of which does not have exactly two
    // children (no builder of the reference's makes one) is walked in the reference's order only.
    // (the reference-order part of an ordered walk reads the record behind the node it stands on as well: one record of padding)
    nodes.push_back(make_float4(0.f, 0.f, 0.f, 0.f)); nodes.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<float4> wide(2 * nodes.size(), make_float4(0.f, 0.f, 0.f, 0.f));
    auto link_pairs = [&](uint32_t base, uint32_t n) -> bool {
        auto word = [&](uint32_t i, int w) { uint32_t v; const float4& q = nodes[2 * (size_t)(base + i) + 1]; std::memcpy(&v, w ? &q.w : &q.z, 4); return v; };
        auto skip_of = [&](uint32_t i) { const uint32_t a = word(i, 0); return (a & LEAF_BIT) ? i + 1u : a; };
        for (uint32_t i = 0; i < n; i++) {
            if (word(i, 0) & LEAF_BIT) continue;
            const uint32_t l = i + 1u;
            if (l >= n) return false;
            const uint32_t r = skip_of(l);
            if (r <= l || r >= n || skip_of(r) != skip_of(i)) return false;
            float4* rec = &wide[4 * (size_t)(base + i)];
            rec[0] = nodes[2 * (size_t)(base + l)]; rec[1] = nodes[2 * (size_t)(base + l) + 1];
            rec[2] = nodes[2 * (size_t)(base + r)]; rec[3] = nodes[2 * (size_t)(base + r) + 1];
        }
        return true;
    };
    bool pairs_meshes = true;
    for (const DevMesh& m : meshes) pairs_meshes = link_pairs(m.node_base, m.n_nodes) && pairs_meshes;
    for (DevTree& t : trees)
        if (!link_pairs(t.node_base, t.n_nodes)) t.deepen &= ~TREE_ORDERED;
    std::vector<DevObject> objects;
    for (uint32_t o = 0; o < sc->n_objects; o++) {
        const rtamd_object& ob = sc->objects[o];
        if (ob.material >= sc->n_materials) return fail(ctx, RTAMD_E_INVALID, "object material index out of bounds");
        if ((uint64_t)ob.op_first + ob.op_count > sc->n_ops) return fail(ctx, RTAMD_E_INVALID, "object op range out of bounds");
        switch (ob.geom_kind) {
            case RTAMD_GEOM_MESH: if (ob.geom_index >= sc->n_meshes) return fail(ctx, RTAMD_E_INVALID, "object mesh index out of bounds"); break;
            case RTAMD_GEOM_CUBOID: if (ob.geom_index >= sc->n_cuboids) return fail(ctx, RTAMD_E_INVALID, "object cuboid index out of bounds"); break;
            case RTAMD_GEOM_TRIANGLE: if (ob.geom_index >= sc->n_triangles) return fail(ctx, RTAMD_E_INVALID, "object triangle index out of bounds"); break;
            case RTAMD_GEOM_UNIT_SPHERE: case RTAMD_GEOM_UNIT_CUBOID: break;
            default: return fail(ctx, RTAMD_E_INVALID, "unknown geometry kind");
        }
        DevObject d{};
        d.geom_kind = ob.geom_kind; d.geom_index = ob.geom_index; d.material = ob.material;
        d.op_first = ob.op_first; d.op_count = ob.op_count;
        if (ob.geom_kind == RTAMD_GEOM_MESH) {
            d.node_base = meshes[ob.geom_index].node_base; d.n_nodes = meshes[ob.geom_index].n_nodes; d.tri_base = meshes[ob.geom_index].tri_base;
            d.normal_base = meshes[ob.geom_index].normal_base;
        }
        // inline wrapper parameters when the chain is a subsequence of Translate, Scale, Rotate
        d.scale = 1.0f; d.inv_scale = 1.0f;
        uint32_t last = 0;
        // Whether a hit can be carried out of the object from the inline parameters too (rotate, scale, then + off): every
        // Translate of the chain, those by +0 included, must stand outside its Scale and Rotate, and adding +0 before the
        // offset must not differ from adding the offset alone (it does for a -0 component: -0 + 0 + -0 = +0, -0 + -0 = -0).
        bool any_translate = false, zero_translate = false, translate_inside = false;
        for (uint32_t k = 0; k < ob.op_count && d.xf_kind != XF_GENERIC; k++) {
            const rtamd_xform_op& op = sc->ops[ob.op_first + k];
            if (op.kind == RTAMD_OP_TRANSLATE) { any_translate = true; translate_inside |= (d.xf_kind & (XF_S | XF_R)) != 0u; }
            // A Translate by (+0, +0, +0) leaves every ray position as it is, bit for bit (x - (+0) == x, also for -0, inf
            // and NaN): the reference's Transform always carries one (transform.rs:224-230), often with a zero offset
            // under another Translate.  It takes no part in entering the object; the hit is still carried out through
            // the full op list.
            if (op.kind == RTAMD_OP_TRANSLATE) {
                uint32_t bits[3];
                std::memcpy(bits, op.v, 12);
                if ((bits[0] | bits[1] | bits[2]) == 0u) { zero_translate = true; continue; }
            }
            const uint32_t bit = op.kind == RTAMD_OP_TRANSLATE ? XF_T : op.kind == RTAMD_OP_SCALE ? XF_S : op.kind == RTAMD_OP_ROTATE ? XF_R : XF_GENERIC;
            if (bit == XF_GENERIC || bit <= last) { d.xf_kind = XF_GENERIC; break; }
            last = bit;
            d.xf_kind |= bit;
            if (bit == XF_T) std::memcpy(d.off, op.v, 12);
            else if (bit == XF_S) { d.scale = op.v[0]; d.inv_scale = op.v[1]; }
            else std::memcpy(d.rot, op.v, 36);
        }
        if (d.xf_kind == XF_GENERIC) { d.scale = 1.0f; d.inv_scale = 1.0f; }
        else {
            bool minus_zero = false;
            for (int c = 0; c < 3; c++) minus_zero |= d.off[c] == 0.0f && std::signbit(d.off[c]);
            if (!translate_inside && !(zero_translate && minus_zero)) d.xf_kind |= XF_OUT_INLINE | (any_translate ? XF_T_OUT : 0u);
        }
        object_box(sc, ob, nodes, meshes, d);
        objects.push_back(d);
    }
    // Instance records and classes (see DevInst): an object whose wrapper chain is inline (at most one Translate, Scale and
    // Rotate each, in that order) belongs to the class of all objects with the same primitive, wrapper kinds and rotation;
    // meshes also share the scale (the class carries the box of the scaled, rotated mesh).  An object without a Translate
    // gets the offset +0, which leaves a position as it is bit for bit.
    std::vector<DevObject> classes;
    std::vector<DevInst> insts(sc->n_objects);
    {
        auto same_class = [](const DevObject& a, const DevObject& b) {
            if (a.geom_kind != b.geom_kind || a.geom_index != b.geom_index || a.xf_kind != b.xf_kind) return false;
            if ((a.xf_kind & XF_R) && std::memcmp(a.rot, b.rot, sizeof(a.rot)) != 0) return false;
            if (a.geom_kind == GEOM_MESH && (std::memcmp(&a.scale, &b.scale, 4) != 0 || std::memcmp(&a.inv_scale, &b.inv_scale, 4) != 0)) return false;
            return true;
        };
        uint32_t last = CLS_NONE;
        for (uint32_t o = 0; o < sc->n_objects; o++) {
            const DevObject& d = objects[o];
            DevInst in{};
            std::memcpy(in.off, d.off, 12);
            in.scale = d.scale; in.inv_scale = d.inv_scale; in.cls = CLS_NONE; in.material = d.material;
            if (!(d.xf_kind & XF_GENERIC)) {
                DevObject key = d;
                key.xf_kind |= XF_T;
                key.material = 0; key.op_first = 0; key.op_count = 0; key.pad1 = 0;
                std::memset(key.off, 0, sizeof(key.off));
                uint32_t c = CLS_NONE;
                if (last != CLS_NONE && same_class(classes[last], key)) c = last;
                for (uint32_t k = 0; k < classes.size() && c == CLS_NONE; k++)
                    if (same_class(classes[k], key)) c = k;
                if (c == CLS_NONE && classes.size() < MAX_CLASSES) {
                    if (key.geom_kind != GEOM_MESH) { key.scale = 1.0f; key.inv_scale = 1.0f; }
                    if (key.xf_kind & XF_BOX) {   // the class box: this object's box with every Translate taken as zero
                        DevObject rel{};
                        object_box(sc, sc->objects[o], nodes, meshes, rel, true);
                        std::memcpy(key.bmin, rel.bmin, 12); std::memcpy(key.bmax, rel.bmax, 12); key.slack = rel.slack;
                    }
                    c = (uint32_t)classes.size();
                    classes.push_back(key);
                }
                if (c != CLS_NONE && (classes[c].xf_kind & XF_BOX)) {
                    // box + offset is rounded on the device: the class box's pad (1e-4 of its size) must dwarf that rounding
                    const DevObject& k = classes[c];
                    for (int a = 0; a < 3 && c != CLS_NONE; a++) {
                        const float reach = std::fabs(d.off[a]) + std::max(std::fabs(k.bmin[a]), std::fabs(k.bmax[a]));
                        const float ulp = std::nextafter(reach, std::numeric_limits<float>::infinity()) - reach;
                        const float pad = 1e-4f * (k.bmax[a] - k.bmin[a]) / 1.0002f;
                        if (!(4.0f * ulp < pad)) c = CLS_NONE;
                    }
                }
                in.cls = c;
                last = c != CLS_NONE ? c : last;
            }
            insts[o] = in;
        }
    }
    std::vector<DevOp> ops;
    for (uint32_t k = 0; k < sc->n_ops; k++) {
        if (sc->ops[k].kind > RTAMD_OP_BOUNDS) return fail(ctx, RTAMD_E_INVALID, "unknown transform op");
        DevOp d{};
        d.kind = sc->ops[k].kind;
        std::memcpy(d.v, sc->ops[k].v, sizeof(d.v));
        ops.push_back(d);
    }
    std::vector<DevRootItem> root;
    for (uint32_t k = 0; k < sc->n_root; k++) {
        const rtamd_root_item& it = sc->root[k];
        if (it.kind == RTAMD_ROOT_OBJECT) { if (it.index >= sc->n_objects) return fail(ctx, RTAMD_E_INVALID, "root object index out of bounds"); }
        else if (it.kind == RTAMD_ROOT_OBJECT_TREE) { if (it.index >= sc->n_trees) return fail(ctx, RTAMD_E_INVALID, "root tree index out of bounds"); }
        else return fail(ctx, RTAMD_E_INVALID, "unknown root item kind");
        root.push_back(DevRootItem{it.kind, it.index});
    }
    std::vector<DevCuboid> cuboids;
    for (uint32_t k = 0; k < sc->n_cuboids; k++) {
        DevCuboid d{};
        std::memcpy(d.center, sc->cuboids[k].center, 12); std::memcpy(d.extent, sc->cuboids[k].extent, 12);
        std::memcpy(d.inv_extent, sc->cuboids[k].inv_extent, 12);
        cuboids.push_back(d);
    }
    std::vector<DevTriangle> triangles;
    for (uint32_t k = 0; k < sc->n_triangles; k++) {
        DevTriangle d{};
        std::memcpy(d.a, sc->triangles[k].a, 12); std::memcpy(d.b, sc->triangles[k].b, 12); std::memcpy(d.c, sc->triangles[k].c, 12);
        triangles.push_back(d);
    }
    std::vector<DevMaterial> materials;
    for (uint32_t k = 0; k < sc->n_materials; k++) {
        const rtamd_material& m = sc->materials[k];
        if (m.kind > RTAMD_MAT_LAMBERT) return fail(ctx, RTAMD_E_INVALID, "unknown material kind");
        if (m.tex_kind > RTAMD_TEX_IMAGE) return fail(ctx, RTAMD_E_INVALID, "unknown texture kind");
        if (m.tex_kind == RTAMD_TEX_IMAGE && (m.tex_param < 0 || (uint32_t)m.tex_param >= sc->n_images || !sc->images))
            return fail(ctx, RTAMD_E_INVALID, "material image index out of bounds");
        DevMaterial d{};
        d.kind = m.kind; d.p0 = m.p0; d.p1 = m.p1; std::memcpy(d.color, m.color, 12);
        d.tex_kind = m.tex_kind; d.tex_param = m.tex_param;
        materials.push_back(d);
    }
    std::vector<DevLight> lights;
    ctx->lights.clear();
    for (uint32_t k = 0; k < sc->n_lights; k++) {
        const rtamd_light& l = sc->lights[k];
        if (l.kind > RTAMD_LIGHT_AO) return fail(ctx, RTAMD_E_INVALID, "unknown light kind");
        DevLight d{};
        d.kind = l.kind; std::memcpy(d.vec, l.vec, 12); d.param = l.param; std::memcpy(d.emission, l.emission, 12);
        lights.push_back(d);
        ctx->lights.push_back(l);
    }
    // skybox: RGB8 -> one RGBA8 word per texel (texture.rs:75-81 reads p[0..2])
    std::vector<uint32_t> sky;
    uint32_t sw = sc->sky_w, sh = sc->sky_h;
    if (sc->skybox_rgb && sw && sh) {
        sky.resize((size_t)sw * sh);
        const uint8_t* p = sc->skybox_rgb;
        for (size_t k = 0; k < sky.size(); k++) sky[k] = (uint32_t)p[3 * k] | ((uint32_t)p[3 * k + 1] << 8) | ((uint32_t)p[3 * k + 2] << 16) | 0xFF000000u;
    } else {
        sw = sh = 1;
        sky.assign(1, 0xFF808080u);
    }
    // image textures: their texels follow the skybox in the same array
    std::vector<DevImage> images;
    for (uint32_t k = 0; k < sc->n_images; k++) {
        const rtamd_image& im = sc->images[k];
        if (!im.rgb || im.w == 0 || im.h == 0) return fail(ctx, RTAMD_E_INVALID, "empty texture image");
        const size_t first = sky.size(), n = (size_t)im.w * im.h;
        if (first + n > 0xFFFFFFF0ull) return fail(ctx, RTAMD_E_INVALID, "texture images too large");
        images.push_back(DevImage{(uint32_t)first, im.w, im.h, 0u});
        sky.resize(first + n);
        for (size_t t = 0; t < n; t++)
            sky[first + t] = (uint32_t)im.rgb[3 * t] | ((uint32_t)im.rgb[3 * t + 1] << 8) | ((uint32_t)im.rgb[3 * t + 2] << 16) | 0xFF000000u;
    }

    DevScene S{};
    if ((rc = upload_vec(ctx, nodes, &S.nodes))) return rc;
    if ((rc = upload_vec(ctx, wide, &S.wide))) return rc;
    if ((rc = upload_vec(ctx, tris, &S.tris))) return rc;
    if ((rc = upload_vec(ctx, normals, &S.normals))) return rc;
    if ((rc = upload_vec(ctx, meshes, &S.meshes))) return rc;
    if ((rc = upload_vec(ctx, trees, &S.trees))) return rc;
    if ((rc = upload_vec(ctx, cuboids, &S.cuboids))) return rc;
    if ((rc = upload_vec(ctx, triangles, &S.triangles))) return rc;
    if ((rc = upload_vec(ctx, ops, &S.ops))) return rc;
    if ((rc = upload_vec(ctx, objects, &S.objects))) return rc;
    if ((rc = upload_vec(ctx, root, &S.root))) return rc;
    if ((rc = upload_vec(ctx, materials, &S.materials))) return rc;
    if ((rc = upload_vec(ctx, lights, &S.lights))) return rc;
    if ((rc = upload_vec(ctx, sky, &S.sky))) return rc;
    if ((rc = upload_vec(ctx, images, &S.images))) return rc;
    if ((rc = upload_vec(ctx, insts, &S.insts))) return rc;
    if ((rc = upload_vec(ctx, classes, &S.classes))) return rc;
    S.n_classes = (uint32_t)classes.size(); S.n_trees = sc->n_trees;
    S.n_root = sc->n_root; S.n_lights = sc->n_lights; S.sky_w = sw; S.sky_h = sh;
    S.n_objects = sc->n_objects; S.n_meshes = sc->n_meshes;
    // LDS staging: of the meshes whose nodes + triangles fit in 52 KiB (two 512-thread blocks per CU), the one that
    // most objects instance, and only if at least a quarter of the mesh objects do (every block pays for the copy)
    S.lds_mesh = MISS; S.lds_bytes = 0;
    {
        std::vector<uint32_t> refs(sc->n_meshes, 0u);
        uint32_t mesh_objects = 0;
        for (const DevObject& o : objects)
            if (o.geom_kind == GEOM_MESH) { refs[o.geom_index]++; mesh_objects++; }
        uint32_t best_refs = 0;
        for (uint32_t m = 0; m < sc->n_meshes; m++) {
            const size_t bytes = (size_t)meshes[m].n_nodes * 32 + (size_t)sc->meshes[m].n_tris * 48;
            if (bytes > 0 && bytes <= 52 * 1024 && refs[m] > best_refs) { S.lds_mesh = m; S.lds_bytes = (uint32_t)bytes; best_refs = refs[m]; }
        }
        if (S.lds_mesh != MISS && (uint64_t)best_refs * 4u < mesh_objects) { S.lds_mesh = MISS; S.lds_bytes = 0; }
        S.lds_sched_bytes = S.lds_mesh != MISS ? lds_mesh_bytes(meshes[S.lds_mesh].n_nodes, sc->meshes[S.lds_mesh].n_tris) : 0u;
    }
    // Scene tables small enough for LDS (objects, ops, cuboids, root list): at most 8 KiB
    S.n_ops = sc->n_ops; S.n_cuboids = sc->n_cuboids;
    S.plain_boxes = plain_boxes ? 1u : 0u;
    S.plain_tree_boxes = plain_tree_boxes ? 1u : 0u;
    S.any_box = 0;
    for (const DevObject& o : objects) S.any_box |= (o.xf_kind & XF_BOX) ? 1u : 0u;
    S.features = 0;   // what the objects need besides mesh walks: picks the lean or the general form of the scheduler kernels
    for (const DevObject& o : objects) {
        if (o.geom_kind == GEOM_UNIT_SPHERE) S.features |= FEAT_SPHERE;
        else if (o.geom_kind == GEOM_CUBOID) S.features |= FEAT_CUBOID;
        else if (o.geom_kind != GEOM_MESH) S.features |= FEAT_OTHER;
        if (o.xf_kind & XF_GENERIC) S.features |= FEAT_GENERIC_XF;
    }
    // a root list of plain objects with consecutive indices is walked as one range, without reading the list
    S.root_first = MISS;
    if (sc->n_root > 0 && sc->root[0].kind == RTAMD_ROOT_OBJECT) {
        bool range = true;
        for (uint32_t k = 0; k < sc->n_root && range; k++)
            range = sc->root[k].kind == RTAMD_ROOT_OBJECT && sc->root[k].index == sc->root[0].index + k;
        if (range) S.root_first = sc->root[0].index;
    }
    {
        size_t tb = (size_t)sc->n_objects * sizeof(DevObject) + (size_t)sc->n_ops * 48 + (size_t)sc->n_cuboids * 48 + (((size_t)sc->n_root * 8 + 15) & ~(size_t)15);
        S.tab_bytes = tb <= 8192 ? (uint32_t)tb : 0u;
    }
    {   // large scenes: classes, cuboids, tree headers and the root list in LDS (at most 8 KiB), instance records in global memory
        size_t mb = classes.size() * sizeof(DevObject) + (size_t)sc->n_cuboids * 48 + (size_t)sc->n_trees * 16 + (((size_t)sc->n_root * 8 + 15) & ~(size_t)15);
        S.mid_bytes = (S.tab_bytes == 0 && !classes.empty() && mb <= 8192) ? (uint32_t)mb : 0u;
        // (beside a staged mesh those kernel instances hold 40 KB of fixed LDS -- ray staging, triangle phase, ordered-walk stack --
        // and two blocks share a CU's 160 KB: mesh + tables must fit the other 40 KB, else the tables stay in global memory)
        if (S.lds_sched_bytes + S.mid_bytes > 40u * 1024u) S.mid_bytes = 0u;
    }
    if (std::getenv("RTAMD_VERBOSE")) {
        size_t classed = 0;
        for (const DevInst& in : insts) classed += in.cls != CLS_NONE;
        std::fprintf(stderr, "rtamd: %u objects (%zu in %zu classes), %u meshes, %zu nodes, %zu triangles; LDS mesh %d (%u B), tables %u B, mid tables %u B\n",
                     sc->n_objects, classed, classes.size(), sc->n_meshes, nodes.size() / 2, tris.size() / 3, (int)S.lds_mesh, S.lds_bytes, S.tab_bytes, S.mid_bytes);
    }
    // Share of a ray queue dealt to the waves up front.  Short, even batches (a handful of objects): 6/8, the rest of the
    // queue would otherwise be bound by the claim atomics (bunny scene: 3.09 Grays/s at 6, 2.95 at 4, 2.78 at 3).  Object trees:
    // batch times spread over a decade, the shared part evens them out (instanced grid 1.97 -> 2.04 Grays/s, 10 000 spheres
    // 2.69 -> 2.85 at 3/8).
    if (!ctx->static_share_set) {
        bool big_tree = false;
        for (const DevTree& t : trees) big_tree = big_tree || t.n_nodes >= 256u;
        ctx->cfg.static_share = big_tree ? 3 : 6;
    }
    S.deepen = ctx->deepen >= 1 ? 1u : 0u;   // what the render entry points pass to their kernels; the batch entry points decide per call
    ctx->ordered_ok = (pairs_meshes ? ORD_MESH : 0u) | ORD_TREE;   // (object trees carry their own flag, DevTree::deepen)
    S.ordered = ctx->ordered >= 1 ? ctx->ordered_ok : 0u;
    S.ordered_cap = ctx->ordered_cap;
    ctx->S = S;
    ctx->have_scene = true;
    return RTAMD_OK;
}

namespace {
// The scene as the batch entry points hand it to their kernels.  They are the parity entry points: every ray gets the
// reference's answer, the spurious t = 0 hits of near-parallel triangles (triangle.rs:26-28) included, wherever along the
// ray such a triangle lies -- which only a walk over everything the ray crosses can find.  The walk in passes of growing
// reach is therefore opt-in here (RTAMD_OPT_TREE_DEEPENING = 2).
DevScene batch_scene(const rtamd_context* ctx) {
    DevScene S = ctx->S;
    S.deepen = ctx->deepen >= 2 ? 1u : 0u;
    S.ordered = ctx->ordered >= 2 ? ctx->ordered_ok : 0u;
    return S;
}
}  // namespace

int rtamd_trace_closest_device(rtamd_context* ctx, const rtamd_rays* rays, size_t n, const rtamd_hits* out, void* stream) {
    if (!ctx || !rays || !out) return RTAMD_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RTAMD_E_NOSCENE, "no scene uploaded");
    if (n > 0x7FFFFFFFull) return fail(ctx, RTAMD_E_INVALID, "at most 2^31-1 rays per call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    RayBatch rb{rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tmax};
    HitBatch hb{out->obj, out->prim, out->t, out->u, out->v};
    uint32_t* head = ctx->work_head + (ctx->head_next++ % HEAD_RING);
    // kernel timing only on the context's own stream, where the events are collected right after the call
    if (!stream) timed_begin(ctx, 0, st);
    HIP_TRY(ctx, launch_trace_closest(batch_scene(ctx), ctx->cfg, rb, hb, n, head, st));
    if (!stream) timed_end(ctx, st);
    if (!stream) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        timed_collect(ctx, nullptr);
    }
    return RTAMD_OK;
}

int rtamd_trace_any_device(rtamd_context* ctx, const rtamd_rays* rays, size_t n, uint8_t* out, void* stream) {
    if (!ctx || !rays || !out) return RTAMD_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RTAMD_E_NOSCENE, "no scene uploaded");
    if (n > 0x7FFFFFFFull) return fail(ctx, RTAMD_E_INVALID, "at most 2^31-1 rays per call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    RayBatch rb{rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tmax};
    uint32_t* head = ctx->work_head + (ctx->head_next++ % HEAD_RING);
    if (!stream) timed_begin(ctx, 1, st);
    HIP_TRY(ctx, launch_trace_any(batch_scene(ctx), ctx->cfg, rb, out, n, head, st));
    if (!stream) timed_end(ctx, st);
    if (!stream) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        timed_collect(ctx, nullptr);
    }
    return RTAMD_OK;
}

// The variant of rtamd_trace_closest_device that allocates nothing: the scheduler kernel's queue head lives in the
// static `scratch` array of this GPU instead of the context's heap allocation.  No hipMalloc / hipFree, no host
// synchronisation: one 4-byte hipMemsetAsync and one kernel launch on the caller's stream.
int rtamd_trace_closest_static(rtamd_context* ctx, const rtamd_rays* rays, size_t n, const rtamd_hits* out, void* stream) {
    if (!ctx || !rays || !out) return RTAMD_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RTAMD_E_NOSCENE, "no scene uploaded");
    if (n > RTAMD_STATIC_MAX_RAYS) return fail(ctx, RTAMD_E_INVALID, "rtamd_trace_closest_static: at most RTAMD_STATIC_MAX_RAYS (2^31-1) rays per call");
    if (n == 0) return RTAMD_OK;
    if (ctx->device < 0 || ctx->device >= MAX_DEVICES) return fail(ctx, RTAMD_E_INVALID, "rtamd_trace_closest_static: device index beyond the library's table");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t* head;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        uint32_t* base = g_scratch_ptr[ctx->device];
        if (!base) return fail(ctx, RTAMD_E_DEVICE, "rtamd_trace_closest_static: the scratch array of this GPU was not resolved");
        head = base + (g_scratch_next[ctx->device]++ % HEAD_RING);
    }
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    RayBatch rb{rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tmax};
    HitBatch hb{out->obj, out->prim, out->t, out->u, out->v};
    LaunchCfg cfg = ctx->cfg;
    cfg.scheduler = 1;   // the kernel that uses the scratch word
    HIP_TRY(ctx, launch_trace_closest(batch_scene(ctx), cfg, rb, hb, n, head, st));   // zeroes its head word on `st` first
    if (!stream) HIP_TRY(ctx, hipStreamSynchronize(st));
    return RTAMD_OK;
}

int rtamd_trace_closest(rtamd_context* ctx, const rtamd_rays* rays, size_t n, const rtamd_hits* out) {
    if (!ctx || !rays || !out) return RTAMD_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RTAMD_E_NOSCENE, "no scene uploaded");
    if (n == 0) return RTAMD_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_batch(ctx, n * 12 * 4);
    if (rc) return rc;
    float* d = (float*)ctx->batch_buf;
    const float* src[7] = {rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tmax};
    for (int k = 0; k < 7; k++) HIP_TRY(ctx, hipMemcpyAsync(d + k * n, src[k], n * 4, hipMemcpyHostToDevice, ctx->stream));
    rtamd_rays dr{d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, d + 5 * n, d + 6 * n};
    rtamd_hits dh{(uint32_t*)(d + 7 * n), (uint32_t*)(d + 8 * n), d + 9 * n, d + 10 * n, d + 11 * n};
    rc = rtamd_trace_closest_device(ctx, &dr, n, &dh, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(out->obj, dh.obj, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out->prim, dh.prim, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out->t, dh.t, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out->u, dh.u, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out->v, dh.v, n * 4, hipMemcpyDeviceToHost));
    return RTAMD_OK;
}

int rtamd_trace_any(rtamd_context* ctx, const rtamd_rays* rays, size_t n, uint8_t* out) {
    if (!ctx || !rays || !out) return RTAMD_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RTAMD_E_NOSCENE, "no scene uploaded");
    if (n == 0) return RTAMD_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_batch(ctx, n * 8 * 4);
    if (rc) return rc;
    float* d = (float*)ctx->batch_buf;
    const float* src[7] = {rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tmax};
    for (int k = 0; k < 7; k++) HIP_TRY(ctx, hipMemcpyAsync(d + k * n, src[k], n * 4, hipMemcpyHostToDevice, ctx->stream));
    rtamd_rays dr{d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, d + 5 * n, d + 6 * n};
    uint8_t* dout = (uint8_t*)(d + 7 * n);
    rc = rtamd_trace_any_device(ctx, &dr, n, dout, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(out, dout, n, hipMemcpyDeviceToHost));
    return RTAMD_OK;
}

namespace {
constexpr size_t PathFrame_WORDS = 20;   // words of a path frame of the depth-first integrators (rt_kernels.hip)
CameraDev make_camera(const rtamd_camera& cam) {
    CameraDev C;
    C.pos = f3{cam.pos[0], cam.pos[1], cam.pos[2]};
    C.dir = f3{cam.dir[0], cam.dir[1], cam.dir[2]};
    C.horizontal = f3{cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]};
    C.vertical = f3{cam.vertical[0], cam.vertical[1], cam.vertical[2]};
    C.width = cam.width; C.height = cam.height;
    return C;
}

// scatter_rays != 1: the path tree of a pixel is walked depth first by one lane (k_pixel_depth_first), which keeps the
// reference's random-stream order; the wavefront pipeline below covers scatter_rays = 1.
int render_depth_first(rtamd_context* ctx, const rtamd_camera* cam, const rtamd_integrator* integ, const rtamd_region* region, size_t total_slots,
                       float* out, rtamd_stats* stats) {
    const int D = integ->max_depth > 0 ? integ->max_depth : 0;
    const size_t cap = std::min(total_slots, std::min(ctx->max_paths, (size_t)1 << 22));   // the stack costs 80 B per slot and depth
    const size_t stack_words = (size_t)(D ? D : 1) * 20 * cap;
    int rc = ensure_batch(ctx, 256 + stack_words * 4);
    if (rc) return rc;
    unsigned long long* dstats = (unsigned long long*)ctx->batch_buf;
    float* stack = (float*)((char*)ctx->batch_buf + 256);
    const RegionDev R = make_region(*region);
    const CameraDev C = make_camera(*cam);
    hipEvent_t ev0 = get_event(ctx), ev1 = get_event(ctx);
    HIP_TRY(ctx, hipMemsetAsync(dstats, 0, 16, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ev0, ctx->stream));
    uint64_t launches = 0;
    for (size_t begin = 0; begin < total_slots; begin += cap) {
        const uint32_t chunk_n = (uint32_t)std::min(cap, total_slots - begin);
        HIP_TRY(ctx, launch_pixel_depth_first(ctx->S, ctx->cfg, R, C, D, integ->shadow_rays, integ->scatter_rays, (uint32_t)integ->antialiasing,
                                              (uint32_t)begin, chunk_n, stack, (uint32_t)cap, out, dstats, ctx->stream));
        launches++;
    }
    HIP_TRY(ctx, hipEventRecord(ev1, ctx->stream));
    unsigned long long h[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(h, dstats, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    ctx->event_pool.push_back(ev0);
    ctx->event_pool.push_back(ev1);
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->closest_rays = h[0];
        stats->any_rays = h[1];
        stats->paths = owned_pixels(*region) * (uint64_t)integ->antialiasing * integ->antialiasing;
        stats->launches = launches;
        stats->device_ms = ms;
    }
    return RTAMD_OK;
}

// scatter_rays != 1 on chunks that fill the GPU: the same per-pixel operation sequence as render_depth_first, run as
// rounds of (one scheduler-kernel launch over every pixel's next ray) + (one step of every pixel's coroutine); see
// k_df_step.  The host reads the number of active pixels back every few rounds to know when the chunk is finished.
int render_depth_first_rounds(rtamd_context* ctx, const rtamd_camera* cam, const rtamd_integrator* integ, const rtamd_region* region, size_t total_slots,
                              uint32_t S_total, uint32_t S_shoot, float* out, rtamd_stats* stats) {
    typedef CounterSlots CS;
    const int D = integ->max_depth;
    // the frame stack costs 80 B per slot and depth: at most 6 GiB of it
    size_t cap = std::min(total_slots, ctx->max_paths);
    cap = std::min(cap, std::max<size_t>(65536, ((size_t)6 << 30) / ((size_t)D * PathFrame_WORDS * 4)));
    const size_t stack_bytes = (size_t)D * PathFrame_WORDS * cap * 4;
    int rc = ensure_batch(ctx, stack_bytes);
    if (rc) return rc;
    if (ctx->ws.empty()) ctx->ws.resize(1);
    rtamd_context::Workspace& ws = ctx->ws[0];
    rc = ensure_wave(ctx, ws, cap, S_total);
    if (rc) return rc;
    WaveBufs W = ws.W;
    W.cap = (uint32_t)ws.cap;
    hipStream_t st = ws.stream;
    // (the frame stack is laid out for `cap` slots, whatever the capacity of the workspace, which an earlier render may have
    // left larger)
    DfFrame F{make_region(*region), make_camera(*cam), (uint32_t)integ->antialiasing, D, integ->shadow_rays, integ->scatter_rays, S_shoot,
              (float*)ctx->batch_buf, out, (uint32_t)cap};
    hipEvent_t ev0 = get_event(ctx), ev1 = get_event(ctx);
    HIP_TRY(ctx, hipEventRecord(ev0, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(st, ev0, 0));
    HIP_TRY(ctx, hipMemsetAsync(W.stats, 0, 32, st));
    uint64_t launches = 0;
    for (size_t begin = 0; begin < total_slots; begin += cap) {
        const uint32_t chunk_n = (uint32_t)std::min(cap, total_slots - begin);
        const size_t upper = (size_t)chunk_n * (1 + S_shoot);
        HIP_TRY(ctx, hipMemsetAsync(W.counts, 0, CS::WORDS * 4, st));
        HIP_TRY(ctx, launch_df_begin(W, F, (uint32_t)begin, chunk_n, 0u, st));
        launches++;
        uint32_t next_check = 8;
        for (uint32_t r = 0;; r++) {
            const uint32_t in = (r % 3u) * CS::TOTAL, nxt = ((r + 1u) % 3u) * CS::TOTAL, zero = ((r + 2u) % 3u) * CS::TOTAL;
            timed_begin(ctx, 0, st);
            HIP_TRY(ctx, launch_wave_trace(ctx->S, ctx->cfg, W, r & 1u, in + CS::Q, MISS, in + CS::ANY, MISS, 0u, 1u, in + CS::HEAD, upper, st));
            timed_end(ctx, st);
            timed_begin(ctx, 2, st);
            HIP_TRY(ctx, launch_df_step(ctx->S, ctx->cfg, W, F, (uint32_t)begin, r & 1u, in, nxt, zero, chunk_n, st));
            timed_end(ctx, st);
            launches += 2;
            if (r + 1u == next_check) {
                uint32_t n_active = 0;
                HIP_TRY(ctx, hipMemcpyAsync(&n_active, W.counts + nxt + CS::Q + 1, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(ctx, hipStreamSynchronize(st));
                if (n_active == 0) break;
                // the last pixels standing leave rounds that are nearly empty: look more often then
                next_check = r + 1u + ((uint64_t)n_active * 64u < chunk_n ? 2u : 8u);
            }
        }
    }
    uint64_t hstats[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(hstats, W.stats, 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipEventRecord(ws.done, st));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ws.done, 0));
    HIP_TRY(ctx, hipEventRecord(ev1, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    ctx->event_pool.push_back(ev0);
    ctx->event_pool.push_back(ev1);
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->closest_rays = hstats[0];
        stats->any_rays = hstats[1];
        stats->paths = owned_pixels(*region) * (uint64_t)integ->antialiasing * integ->antialiasing;
        stats->launches = launches;
        stats->device_ms = ms;
    }
    timed_collect(ctx, stats);
    return RTAMD_OK;
}

// scatter_rays = 1 with every wave owning a range of pixels from their first camera ray on (rt_mega.hip): per chunk of
// 256 * waves pixel slots one seeding launch, then one launch per window of the pixels' samples: [0, 1) over ranges of equal
// length, then [1, 4), [4, 16), [16, ...) (as many as RTAMD_OPT_FRAME_REBALANCE says), each with the units of 64 pixel slots dealt
// to the waves for equal cost by what the window before it measured.
static constexpr size_t FRAME_KERNEL_AUTO_SLOTS = 1310720;   // 1.25 Mi pixel slots
int render_frame_kernel(rtamd_context* ctx, const rtamd_camera* cam, const rtamd_integrator* integ, const rtamd_region* region, size_t total_slots,
                        uint32_t S_total, uint32_t S_shoot, float* out, rtamd_stats* stats) {
    const uint32_t n_waves = mega_waves(ctx->S, ctx->cfg);
    // (a chunk leaves every wave one bank of its pool free on average, for the deal by cost)
    const size_t cap = (size_t)n_waves * 64u * (mega_banks() - 1u);
    if (ctx->ws.empty()) ctx->ws.resize(1);
    rtamd_context::Workspace& ws = ctx->ws[0];
    int rc = ensure_wave(ctx, ws, cap, S_total);
    if (rc) return rc;
    WaveBufs W = ws.W;
    W.cap = (uint32_t)ws.cap;
    hipStream_t st = ws.stream;
    // one scratch buffer: [units of the waves | owner of every unit | cost of every unit | 2 words per wave | path frames]
    const size_t range_words = (size_t)n_waves * mega_banks(), group_words = cap / 64 + 1;
    const bool tree = integ->scatter_rays != 1;
    const size_t stack_words = tree ? (size_t)integ->max_depth * 20 * ws.cap : 0;
    rc = ensure_batch(ctx, (range_words + 2 * group_words + 2 * (size_t)n_waves + stack_words) * 4);
    if (rc) return rc;
    uint32_t* buf = (uint32_t*)ctx->batch_buf;
    static const bool frame_squares = std::getenv("RTAMD_FRAME_SQUARES") != nullptr;   // (experiments: the frame kernel on 8 x 8 squares)
    const MegaFrame F{make_region(*region, frame_squares), make_camera(*cam), (uint32_t)integ->antialiasing, integ->max_depth, integ->shadow_rays, S_shoot,
                      (uint32_t)ctx->frame_step_min, out, buf, buf + range_words, buf + range_words + group_words,
                      integ->scatter_rays, (float*)(buf + range_words + 2 * group_words + 2 * (size_t)n_waves),
                      buf + range_words + 2 * group_words};
    const uint32_t n_samples = (uint32_t)integ->antialiasing * (uint32_t)integ->antialiasing;
    // windows of samples: [0, 1) [1, w1) [w1, n); few samples, or rebalancing off: one window
    std::vector<uint32_t> ends;
    if (n_samples >= (tree ? 4u : 8u))
        for (uint32_t k = 0, e = 1; k < (uint32_t)ctx->frame_rebalance && e * 2u <= n_samples; k++, e *= 4u) ends.push_back(e);
    ends.push_back(n_samples);
    hipEvent_t ev0 = get_event(ctx), ev1 = get_event(ctx);
    HIP_TRY(ctx, hipEventRecord(ev0, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(st, ev0, 0));
    HIP_TRY(ctx, hipMemsetAsync(W.stats, 0, 32, st));
    uint64_t launches = 0;
    std::vector<uint32_t> units(range_words), owner(group_words);
    std::vector<float> costs(group_words);
    for (size_t begin = 0; begin < total_slots; begin += cap) {
        const uint32_t chunk_n = (uint32_t)std::min(cap, total_slots - begin);
        HIP_TRY(ctx, launch_mega_init(ctx->S, ctx->cfg, W, F, (uint32_t)begin, chunk_n, st));
        launches++;
        uint32_t s_begin = 0;
        for (size_t k = 0; k < ends.size(); k++) {
            if (k > 0) {   // ranges of equal cost by the rays counted so far
                HIP_TRY(ctx, launch_mega_group_cost(ctx->S, ctx->cfg, W, F, chunk_n, st));
                launches++;
                HIP_TRY(ctx, hipMemcpyAsync(costs.data(), F.group_cost, ((size_t)chunk_n + 63) / 64 * 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(ctx, hipStreamSynchronize(st));
            }
            mega_assign(k > 0 ? costs.data() : nullptr, chunk_n, n_waves, units.data(), owner.data());
            HIP_TRY(ctx, hipMemcpyAsync(F.units, units.data(), range_words * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(ctx, hipMemcpyAsync(F.owner, owner.data(), ((size_t)chunk_n + 63) / 64 * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));   // (the table is reused for the next window)
            timed_begin(ctx, 0, st);
            HIP_TRY(ctx, launch_mega(ctx->S, ctx->cfg, W, F, (uint32_t)begin, chunk_n, s_begin, ends[k], st));
            timed_end(ctx, st);
            launches++;
            if (std::getenv("RTAMD_FRAME_DEBUG")) {   // how even the window's waves were
                std::vector<uint32_t> ticks(n_waves);
                HIP_TRY(ctx, hipMemcpyAsync(ticks.data(), F.wave_scale + n_waves, (size_t)n_waves * 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(ctx, hipStreamSynchronize(st));
                double sum = 0; uint32_t mx = 0, used = 0, most = 0, argmax = 0;
                auto held = [&](uint32_t w) { uint32_t n = 0; for (uint32_t q = 0; q < mega_banks(); q++) n += units[(size_t)w * mega_banks() + q] != 0xFFFFFFFFu; return n; };
                for (uint32_t w = 0; w < n_waves; w++) {
                    const uint32_t n = held(w);
                    if (!n) continue;
                    used++; sum += ticks[w];
                    if (ticks[w] > mx) { mx = ticks[w]; argmax = w; }
                    most = std::max(most, n);
                }
                {   // by the wave's place in its block (wave numbering runs over the blocks first)
                    const uint32_t per = n_waves / 8;
                    std::fprintf(stderr, "  mean ticks by wave of the block:");
                    for (uint32_t k = 0; k < 8; k++) { double t = 0; for (uint32_t b = 0; b < per; b++) t += ticks[k * per + b]; std::fprintf(stderr, " %.0f", t / per); }
                    std::fprintf(stderr, "\n");
                }
                std::fprintf(stderr, "frame kernel: samples [%u, %u): %u waves with pixels, ticks mean %.0f max %u (x%.2f; wave %u with %u units), most units %u\n",
                             s_begin, ends[k], used, sum / std::max(1u, used), mx, mx / (sum / std::max(1u, used)), argmax, held(argmax), most);
            }
            s_begin = ends[k];
        }
    }
    uint64_t hstats[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(hstats, W.stats, 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipEventRecord(ws.done, st));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ws.done, 0));
    HIP_TRY(ctx, hipEventRecord(ev1, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    ctx->event_pool.push_back(ev0);
    ctx->event_pool.push_back(ev1);
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->closest_rays = hstats[0];
        stats->any_rays = hstats[1];
        stats->paths = owned_pixels(*region) * (uint64_t)integ->antialiasing * integ->antialiasing;
        stats->launches = launches;
        stats->device_ms = ms;
    }
    timed_collect(ctx, stats);
    // (1: a wave gave up after its step limit, 2: a wave ended with pixels unfinished, 3 / 4: a pixel or hit record out of range, 5: a
    // range longer than a wave's pool)
    if (hstats[2] != 0) return fail(ctx, RTAMD_E_DEVICE, "frame kernel: internal error " + std::to_string(hstats[2]));
    return RTAMD_OK;
}
}  // namespace

size_t rtamd_region_pixels(const rtamd_region* region) {
    if (!region_ok(region)) return 0;
    return owned_tiles(*region) * (size_t)region->tile_w * region->tile_h;
}

int rtamd_render_device(rtamd_context* ctx, const rtamd_camera* cam, const rtamd_integrator* integ, const rtamd_region* region,
                        float* out, rtamd_stats* stats) {
    if (!ctx || !cam || !integ || !out) return RTAMD_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RTAMD_E_NOSCENE, "no scene uploaded");
    if (!region_ok(region)) return fail(ctx, RTAMD_E_INVALID, "bad region");
    if (region->x0 + (uint64_t)region->w > cam->width || region->y0 + (uint64_t)region->h > cam->height)
        return fail(ctx, RTAMD_E_INVALID, "region exceeds the camera image");
    if (integ->antialiasing < 1) return fail(ctx, RTAMD_E_INVALID, "antialiasing must be >= 1");
    if (integ->shadow_rays < 1) return fail(ctx, RTAMD_E_INVALID, "shadow_rays must be >= 1");
    if (integ->max_depth > (int)CounterSlots::MAX_DEPTH) return fail(ctx, RTAMD_E_UNSUPPORTED, "max_depth above 64 is not supported");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    uint32_t S_total = 0;
    uint32_t S_shoot = 0;   // of those, the samples that can need a shadow ray
    for (const auto& l : ctx->lights) {
        const uint32_t ns = light_samples(l, integ->shadow_rays);
        S_total += ns;
        if (!(l.kind == RTAMD_LIGHT_AMBIENT || (l.kind == RTAMD_LIGHT_AO && l.param == 0.0f))) S_shoot += ns;
    }
    const size_t total_slots = rtamd_region_pixels(region);
    if (total_slots == 0) return RTAMD_OK;
    if (total_slots > 0xFFFFFFF0ull) return fail(ctx, RTAMD_E_INVALID, "region too large");
    if (integ->scatter_rays != 1) {
        // the round pipeline needs a ray per pixel to start with (max_depth >= 1); RTAMD_OPT_DEPTH_FIRST_ROUNDS = 0 selects
        // the one-lane-per-pixel kernel
        // (measured, rt_mega.hip: branching scatter rays run 1.2-1.5 times faster in the frame kernel than in rounds of launches
        // while a frame is about one chunk of it; larger frames fill the rounds' launches and the rounds win)
        const bool pays = ctx->frame_kernel == 1 || (ctx->frame_kernel == 2 && total_slots <= FRAME_KERNEL_AUTO_SLOTS);
        if (pays && integ->max_depth >= 1 && integ->max_depth <= 255 && S_shoot <= 7 && mega_supported(ctx->S, ctx->cfg))
            return render_frame_kernel(ctx, cam, integ, region, total_slots, S_total, S_shoot, out, stats);
        if (ctx->df_rounds && integ->max_depth >= 1) return render_depth_first_rounds(ctx, cam, integ, region, total_slots, S_total, S_shoot, out, stats);
        return render_depth_first(ctx, cam, integ, region, total_slots, out, stats);
    }
    if (ctx->frame_kernel == 1 && integ->max_depth >= 1 && S_shoot <= 7 && mega_supported(ctx->S, ctx->cfg))   // (on request only: the pipeline is faster)
        return render_frame_kernel(ctx, cam, integ, region, total_slots, S_total, S_shoot, out, stats);
    // pixel groups: at most n_streams in flight, none smaller than 64 Ki pixels unless the region is
    const size_t G = (size_t)ctx->n_streams;
    size_t cap = (total_slots + G - 1) / G;
    cap = std::max<size_t>(cap, 65536);
    cap = (cap + 1023) / 1024 * 1024;
    cap = std::min(std::min(cap, total_slots), ctx->max_paths);
    const size_t n_groups = (total_slots + cap - 1) / cap;
    const size_t n_ws = std::min(G, n_groups);
    if (ctx->ws.size() < n_ws) ctx->ws.resize(n_ws);
    for (size_t w = 0; w < n_ws; w++) {
        int rc = ensure_wave(ctx, ctx->ws[w], cap, S_total);
        if (rc) return rc;
    }
    RegionDev R = make_region(*region);
    const CameraDev C = make_camera(*cam);

    static const bool host_timing = std::getenv("RTAMD_HOST_TIMING") != nullptr;
    // the resolve of bounce d - 1 and the shading of bounce d as one launch (0: two launches, one after the other)
    static const bool merge_resolve = !(std::getenv("RTAMD_MERGE_RESOLVE") && std::atoi(std::getenv("RTAMD_MERGE_RESOLVE")) == 0);
    const auto ht0 = std::chrono::steady_clock::now();
    hipEvent_t ev0 = get_event(ctx), ev1 = get_event(ctx);
    HIP_TRY(ctx, hipEventRecord(ev0, ctx->stream));
    for (size_t w = 0; w < n_ws; w++) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->ws[w].stream, ev0, 0));
        HIP_TRY(ctx, hipMemsetAsync(ctx->ws[w].W.stats, 0, 32, ctx->ws[w].stream));
    }
    uint64_t launches = 0;
    const uint32_t aa = (uint32_t)integ->antialiasing;
    const int D = integ->max_depth;
    typedef CounterSlots CS;
    // The sample passes of the groups are enqueued round-robin so that the host feeds every stream
    // from the start instead of finishing one group's command list first.
    for (size_t g0 = 0; g0 < n_groups; g0 += n_ws) {
        const size_t g1 = std::min(n_groups, g0 + n_ws);
        for (uint32_t pass = 0; pass <= aa * aa; pass++) {
            for (size_t g = g0; g < g1; g++) {
                rtamd_context::Workspace& ws = ctx->ws[g - g0];
                WaveBufs W = ws.W;
                W.cap = (uint32_t)ws.cap;
                hipStream_t st = ws.stream;
                const size_t begin = g * cap;
                const uint32_t chunk_n = (uint32_t)std::min(cap, total_slots - begin);
                const size_t any_upper = (size_t)chunk_n * (S_total ? S_total : 1);
                const uint32_t n_pass = aa * aa;
                const int sr = integ->shadow_rays;
                // Sample pass k works in counter set k % 3 and radiance buffers k & 1; bounce d of pass k leaves its lit queue and
                // shadow rays in shade set 2 * (k & 1) + (d & 1).  Two kinds of rays do not get a trace launch of their own: the
                // shadow rays of a pass's last bounce ride with the first trace launch of the next pass (after the last pass: a
                // launch of their own) and are resolved into the buffers k & 1 while pass k + 1 fills the other ones; and, for
                // max_depth >= 2, the primary rays of pass k + 1, which depend on nothing that pass k computes, ride with the
                // last bounce of pass k.  The lit hits of bounce d - 1 are resolved by the launch that shades bounce d.
                const uint32_t cb = (pass % 3u) * CS::TOTAL, pb = ((pass + 2u) % 3u) * CS::TOTAL, nb = ((pass + 1u) % 3u) * CS::TOTAL;
                auto sset = [](uint32_t p, int d) -> uint32_t { return 2u * (p & 1u) + ((uint32_t)d & 1u); };
                WaveBufs Wp = W;                        // the previous pass's view: its radiance buffers
                W.L = W.Lb[pass & 1u]; W.Lp = W.Lpb[pass & 1u];
                Wp.L = W.Lb[(pass + 1u) & 1u]; Wp.Lp = W.Lpb[(pass + 1u) & 1u];
                // the lit hits of the previous pass's last bounce (the end of its samples: a skybox term may be pending)
                const ResolveJob carried{Wp.L, Wp.Lp, pb + CS::LIT + (uint32_t)(D > 0 ? D - 1 : 0), sset(pass + 1u, D > 0 ? D - 1 : 0), 1u};
                if (pass == n_pass) {
                    if (D > 0 && n_pass > 0) {  // shadow rays of the last bounce of the last pass
                        timed_begin(ctx, 1, st);
                        HIP_TRY(ctx, launch_wave_trace(ctx->S, ctx->cfg, Wp, 0, MISS, MISS, pb + CS::ANY + (D - 1), MISS, carried.set, 0u, pb + CS::HEAD + D, any_upper, st));
                        timed_end(ctx, st);
                        HIP_TRY(ctx, launch_wave_resolve(ctx->S, ctx->cfg, Wp, sr, carried, chunk_n, st));
                        launches += 2;
                    }
                    // unfolded so far: pass n_pass - 2 (in the buffers this index would use) and pass n_pass - 1
                    HIP_TRY(ctx, launch_finalize(W, R, (uint32_t)begin, chunk_n, (float)n_pass, n_pass >= 2 ? W.L : nullptr, W.Lp, Wp.L, Wp.Lp, out, st));
                    launches++;
                    continue;
                }
                const uint32_t fold = pass >= 2 ? 2u : 1u;
                // max_depth >= 2: the primary queue is built once (pass 0) and kept; the shade launch of bounce 0 rewrites its
                // ray directions for the next pass and zeroes the next pass's counter set, so a pass launches neither a
                // ray-generation kernel nor a memset
                const bool kept = D >= 2;
                const uint32_t pq = kept ? CS::PQ_KEPT : cb + CS::PQ;
                NextPass next{};
                next.zero_base = MISS;
                if (kept && pass + 1 < n_pass) {
                    next.enabled = 1u; next.si = (pass + 1) / aa; next.sj = (pass + 1) % aa; next.aa = aa;
                    next.zero_base = nb;
                    next.cam = C;
                }
                const NextPass none{0u, 0u, 0u, 0u, MISS, C};
                if (pass == 0 || D < 2) {  // otherwise the pass before has generated and traced this pass's primary rays
                    HIP_TRY(ctx, hipMemsetAsync(W.counts + (kept ? 0u : cb), 0, (kept ? CS::WORDS : CS::TOTAL) * 4, st));
                    HIP_TRY(ctx, launch_raygen(W, R, C, (uint32_t)begin, chunk_n, pass / aa, pass % aa, aa, pass == 0, pq, st));
                    launches++;
                    if (D <= 0) {  // integrate_recur returns the skybox immediately (integrator.rs:59-61)
                        HIP_TRY(ctx, launch_mark_miss(W, cb + CS::PQ, chunk_n, st));
                        HIP_TRY(ctx, launch_wave_shade(ctx->S, ctx->cfg, W, sset(pass, 0), 0, 1, sr, fold, cb + CS::PQ, cb + CS::Q + 1, cb + CS::LIT + 0, cb + CS::ANY + 0, S_shoot,
                                                       none, chunk_n, nullptr, nullptr, st));
                        launches += 2;
                        continue;
                    }
                    // primary rays; with max_depth = 1 also the shadow rays of the previous pass (its only bounce)
                    const bool carry = D == 1 && pass > 0;
                    timed_begin(ctx, 0, st);
                    HIP_TRY(ctx, launch_wave_trace(ctx->S, ctx->cfg, W, 0, MISS, pq, MISS, carry ? pb + CS::ANY + 0 : MISS, 0u, carried.set, cb + CS::HEAD + 0,
                                                   chunk_n + (carry ? any_upper : 0), st));
                    timed_end(ctx, st);
                    launches++;
                    if (carry) {
                        HIP_TRY(ctx, launch_wave_resolve(ctx->S, ctx->cfg, Wp, sr, carried, chunk_n, st));
                        launches++;
                    }
                }
                timed_begin(ctx, 2, st);
                HIP_TRY(ctx, launch_wave_shade(ctx->S, ctx->cfg, W, sset(pass, 0), 0, D, sr, fold, pq, cb + CS::Q + 1, cb + CS::LIT + 0, cb + CS::ANY + 0, S_shoot, next, chunk_n,
                                               nullptr, nullptr, st));
                timed_end(ctx, st);
                launches++;
                for (int d = 1; d < D; d++) {
                    const bool ahead = d == D - 1 && pass + 1 < n_pass;   // the next pass's primary rays ride with this launch
                    // closest-hit rays of bounce d with the shadow rays of the bounce before; bounce 1 also takes the shadow
                    // rays that the previous pass's last bounce left
                    const bool carry = d == 1 && pass > 0;
                    const uint32_t a0 = carry ? pb + CS::ANY + (D - 1) : MISS;
                    const uint32_t a1 = cb + CS::ANY + (d - 1);
                    timed_begin(ctx, 0, st);
                    HIP_TRY(ctx, launch_wave_trace(ctx->S, ctx->cfg, W, (uint32_t)(d & 1), cb + CS::Q + d, ahead ? pq : MISS, a0, a1, carried.set, sset(pass, d - 1),
                                                   cb + CS::HEAD + d, chunk_n + (ahead ? chunk_n : 0) + (a0 != MISS ? any_upper : 0) + any_upper, st));
                    timed_end(ctx, st);
                    launches++;
                    timed_begin(ctx, 2, st);
                    const ResolveJob before{W.L, W.Lp, cb + CS::LIT + (uint32_t)(d - 1), sset(pass, d - 1), 0u};
                    if (merge_resolve) {
                        HIP_TRY(ctx, launch_wave_shade(ctx->S, ctx->cfg, W, sset(pass, d), d, D, sr, 0u, cb + CS::Q + d, cb + CS::Q + d + 1, cb + CS::LIT + d, cb + CS::ANY + d, S_shoot,
                                                       none, chunk_n, carry ? &carried : nullptr, &before, st));
                        launches++;
                    } else {
                        if (carry) {
                            HIP_TRY(ctx, launch_wave_resolve(ctx->S, ctx->cfg, Wp, sr, carried, chunk_n, st));
                            launches++;
                        }
                        HIP_TRY(ctx, launch_wave_resolve(ctx->S, ctx->cfg, W, sr, before, chunk_n, st));
                        HIP_TRY(ctx, launch_wave_shade(ctx->S, ctx->cfg, W, sset(pass, d), d, D, sr, 0u, cb + CS::Q + d, cb + CS::Q + d + 1, cb + CS::LIT + d, cb + CS::ANY + d, S_shoot,
                                                       none, chunk_n, nullptr, nullptr, st));
                        launches += 2;
                    }
                    timed_end(ctx, st);
                }
            }
        }
    }
    const auto ht1 = std::chrono::steady_clock::now();
    std::vector<uint64_t> hstats(4 * n_ws, 0);
    for (size_t w = 0; w < n_ws; w++) {
        HIP_TRY(ctx, hipMemcpyAsync(&hstats[4 * w], ctx->ws[w].W.stats, 32, hipMemcpyDeviceToHost, ctx->ws[w].stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ws[w].done, ctx->ws[w].stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ws[w].done, 0));
    }
    const auto ht2 = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipEventRecord(ev1, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const auto ht3 = std::chrono::steady_clock::now();
    for (size_t w = 1; w < n_ws; w++) { hstats[0] += hstats[4 * w]; hstats[1] += hstats[4 * w + 1]; }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    if (host_timing) {
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "rtamd host timing: submit %.0f us, stats copy %.0f us, record + sync %.0f us, total %.0f us, device %.0f us\n", us(ht0, ht1), us(ht1, ht2),
                     us(ht2, ht3), us(ht0, ht3), ms * 1e3);
    }
    ctx->event_pool.push_back(ev0);
    ctx->event_pool.push_back(ev1);
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->closest_rays = hstats[0];
        stats->any_rays = hstats[1];
        stats->paths = owned_pixels(*region) * (uint64_t)aa * aa;
        stats->launches = launches;
        stats->device_ms = ms;
    }
    timed_collect(ctx, stats);
    return RTAMD_OK;
}

int rtamd_render(rtamd_context* ctx, const rtamd_camera* cam, const rtamd_integrator* integ, const rtamd_region* region, float* out,
                 rtamd_stats* stats) {
    if (!ctx || !out) return RTAMD_E_INVALID;
    size_t px = rtamd_region_pixels(region);
    if (px == 0) return fail(ctx, RTAMD_E_INVALID, "bad region");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, px * 12));
    int rc = rtamd_render_device(ctx, cam, integ, region, d, stats);
    if (rc == RTAMD_OK) {
        hipError_t e = hipMemcpy(out, d, px * 12, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, RTAMD_E_DEVICE, hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}

int rtamd_render_multi(rtamd_context* const* ctxs, int n, const rtamd_camera* cam, const rtamd_integrator* integ, uint32_t x0, uint32_t y0,
                       uint32_t w, uint32_t h, uint32_t tile_w, uint32_t tile_h, float* out, rtamd_stats* stats) {
    if (!ctxs || n < 1 || !cam || !integ || !out) return RTAMD_E_INVALID;
    for (int k = 0; k < n; k++)
        if (!ctxs[k]) return RTAMD_E_INVALID;
    if (w == 0 || h == 0 || tile_w == 0 || tile_h == 0) return fail(ctxs[0], RTAMD_E_INVALID, "bad region");
    const uint32_t tiles_x = (w + tile_w - 1) / tile_w, tiles_y = (h + tile_h - 1) / tile_h;
    const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y;
    std::vector<int> rc((size_t)n, RTAMD_OK);
    std::vector<rtamd_stats> st((size_t)n);
    auto work = [&](int k) {
        std::memset(&st[k], 0, sizeof(rtamd_stats));
        if ((uint64_t)k >= n_tiles) return;   // more contexts than tiles
        const rtamd_region reg{x0, y0, w, h, tile_w, tile_h, (uint32_t)k, (uint32_t)n};
        const size_t px = rtamd_region_pixels(&reg);
        std::vector<float> buf;
        try { buf.resize(px * 3); } catch (...) { rc[k] = fail(ctxs[k], RTAMD_E_INVALID, "out of host memory for the tile buffer"); return; }
        rc[k] = rtamd_render(ctxs[k], cam, integ, &reg, buf.data(), &st[k]);
        if (rc[k] != RTAMD_OK) return;
        // tile q of this context is tile k + q * n of the rectangle: copy its rows into the row-major frame
        size_t q = 0;
        for (uint64_t t = (uint64_t)k; t < n_tiles; t += (uint64_t)n, q++) {
            const uint32_t ty = (uint32_t)(t / tiles_x), tx = (uint32_t)(t % tiles_x);
            const uint32_t px0 = tx * tile_w, py0 = ty * tile_h;
            const uint32_t cw = std::min(tile_w, w - px0), ch = std::min(tile_h, h - py0);
            const float* src = buf.data() + q * (size_t)tile_w * tile_h * 3;
            for (uint32_t r = 0; r < ch; r++)
                std::memcpy(out + ((size_t)(py0 + r) * w + px0) * 3, src + (size_t)r * tile_w * 3, (size_t)cw * 12);
        }
    };
    if (n == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        th.reserve((size_t)n);
        for (int k = 0; k < n; k++) th.emplace_back(work, k);
        for (auto& t : th) t.join();
    }
    for (int k = 0; k < n; k++)
        if (rc[k] != RTAMD_OK) {
            if (k != 0) ctxs[0]->error = "context " + std::to_string(k) + ": " + ctxs[k]->error;
            return rc[k];
        }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (int k = 0; k < n; k++) {
            stats->closest_rays += st[k].closest_rays; stats->any_rays += st[k].any_rays; stats->paths += st[k].paths; stats->launches += st[k].launches;
            stats->device_ms = std::max(stats->device_ms, st[k].device_ms);
            stats->trace_closest_ms = std::max(stats->trace_closest_ms, st[k].trace_closest_ms);
            stats->trace_any_ms = std::max(stats->trace_any_ms, st[k].trace_any_ms);
            stats->shade_ms = std::max(stats->shade_ms, st[k].shade_ms);
        }
    }
    return RTAMD_OK;
}

int rtamd_build_bvh(rtamd_context* ctx, const float* boxes, uint32_t n, float hit_cost, rtamd_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* order_out) {
    if (!ctx || !boxes || !nodes_out || !n_nodes_out || !order_out) return RTAMD_E_INVALID;
    if (n == 0 || n > 0x3FFFFFFFu) return fail(ctx, RTAMD_E_INVALID, "rtamd_build_bvh: between 1 and 2^30 - 1 boxes");
    for (size_t k = 0; k < (size_t)n * 6; k++)
        if (!std::isfinite(boxes[k])) return fail(ctx, RTAMD_E_INVALID, "rtamd_build_bvh: boxes must be finite");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float* d_boxes = nullptr;
    rtamd_bvh_node* d_nodes = nullptr;
    uint32_t* d_order = nullptr;
    int rc = RTAMD_OK;
    auto release = [&]() { if (d_boxes) (void)hipFree(d_boxes); if (d_nodes) (void)hipFree(d_nodes); if (d_order) (void)hipFree(d_order); };
#define BUILD_TRY(expr)                                                                                          \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) { rc = fail(ctx, RTAMD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); release(); return rc; } \
    } while (0)
    BUILD_TRY(hipMalloc((void**)&d_boxes, (size_t)n * 24));
    BUILD_TRY(hipMalloc((void**)&d_nodes, (size_t)2 * n * sizeof(rtamd_bvh_node) + sizeof(rtamd_bvh_node)));
    BUILD_TRY(hipMalloc((void**)&d_order, (size_t)n * 4));
    BUILD_TRY(hipMemcpyAsync(d_boxes, boxes, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    uint32_t n_nodes = 0;
    BUILD_TRY(build_bvh_device(d_boxes, n, hit_cost, d_nodes, &n_nodes, d_order, ctx->cfg.num_cus, ctx->stream));
    BUILD_TRY(hipMemcpy(nodes_out, d_nodes, (size_t)n_nodes * sizeof(rtamd_bvh_node), hipMemcpyDeviceToHost));
    BUILD_TRY(hipMemcpy(order_out, d_order, (size_t)n * 4, hipMemcpyDeviceToHost));
#undef BUILD_TRY
    *n_nodes_out = n_nodes;
    release();
    return RTAMD_OK;
}

int rtamd_quantize_device(rtamd_context* ctx, const float* rgb, size_t n, uint8_t* out) {
    if (!ctx || !rgb || !out) return RTAMD_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_quantize(rgb, n, out, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTAMD_OK;
}

#ifdef RTAMD_TIMELINE
// Timeline build only: start / dry / end stamps of the scheduler waves since the last reset (see rt_sched.h).
int rtamd_timeline_read(rtamd_context* ctx, unsigned long long out[8], int reset) {
    if (!ctx) return RTAMD_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, rtamd::timeline_read(out, reset));
    return RTAMD_OK;
}
// The per-launch table: 16 launch slots x 8192 waves x {start, dry, end, trips}, `words` = capacity of `out`.
int rtamd_timeline_dump(rtamd_context* ctx, unsigned long long* out, size_t words, int reset) {
    if (!ctx || !out) return RTAMD_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, rtamd::timeline_dump(out, words, reset));
    return RTAMD_OK;
}
#endif
#ifdef RTAMD_DIAG
// Diagnostic build only: per-phase totals of the scheduler kernels (cycles, steps, active lanes) x 5 phases.
int rtamd_diag_read(rtamd_context* ctx, unsigned long long out[32], int reset) {
    if (!ctx) return RTAMD_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, rtamd::diag_read(out, reset));
    return RTAMD_OK;
}
int rtamd_diag_read_frame(rtamd_context* ctx, unsigned long long out[48], int reset) {
    if (!ctx) return RTAMD_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, rtamd::mega_diag_read(out, reset));
    return RTAMD_OK;
}
// Ticks (10 ns) along the shade kernel's dependency chain as seen by its first thread: out[k] summed over launches,
// out[16 + k] the launch count; k = 0 loads + classification, 2 lit-queue append, 3 light samples, 5 scatter, 7 path-queue append.
int rtamd_diag_read_shade(rtamd_context* ctx, unsigned long long out[32], int reset) {
    if (!ctx) return RTAMD_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, rtamd::diag_read_shade(out, reset));
    return RTAMD_OK;
}
#endif

}  // extern "C"
