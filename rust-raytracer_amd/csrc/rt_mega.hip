// The integrator as ONE launch per chunk of the frame (scatter_rays = 1): every wave owns up to 256 pixels and runs them
// from their first camera ray to their last sample -- tracing (the phase-voting scheduler of rt_sched.h, fed from the
// wave's own pool of ready rays) and shading (pixel_step below, run for up to 64 of the wave's pixels at a time whenever
// their rays have come back).  The chain of dependent launches of the wavefront pipeline (rt_kernels.hip: per sample and
// bounce a trace, a resolve and a shade launch, each as long as its slowest ray) is gone: a pixel's next ray starts as
// soon as the wave gets to it, whatever the other pixels do.  Per pixel the operations, their order and the random
// stream are those of the pipeline (integrate_recur, src/integrator.rs:56-104, with the radiance accumulated along the
// path), so the two produce the same pixels bit for bit.
#include "rt_kernels.h"
#include "rt_sched.h"
#include "rt_stage.h"
#include "rt_integrate.h"
#include <algorithm>
#include <queue>
#include <vector>

namespace rtamd {

// A wave owns up to five units of 64 consecutive pixel slots of the chunk (neighbouring pixels: rays that walk the scene alike
// keep a wave's lanes in the same phases), named by a table.  The units are dealt for equal cost, not equal number: a launch
// covers a window of the pixels' samples, counts the rays of every pixel and clocks every wave, and the host deals the next
// window's units by those costs (launch_mega / mega_assign), because a wave that owned 256 pixels of sky would be done in a
// quarter of the time of one that owned 256 pixels of the model -- and a wave that owned nothing but pixels of the model
// would be bound by their paths' chains of dependent rays whatever their number, so every wave gets a mix.
//
// Per-pixel state in HBM, indexed by the pixel's slot j of the chunk, component-major in the arrays of WaveBufs:
//   rng_lo / rng_hi, sum, Lb[0] (radiance of the sample in work), beta, q_ray[0] (the closest-hit ray in flight),
//   h_rec (its result), p_pix (pixel coordinates), p_slot (rays traced for the pixel so far), q_slot[0] (the word below),
//   set[0]: lit_f, smp (the hit whose light samples are in flight), a_ray / a_vidx (shadow ray k at k * cap + j), vis.
// State word: bits 0-7 depth of the ray in flight, bits 16-31 the sample it belongs to, and
enum : uint32_t {
    PM_RAY = 0x100u,    // a closest-hit ray is in flight
    PM_LIT = 0x200u,    // a lit hit waits for the results of its shadow rays
    PM_NEWS = 0x400u,   // the ray in flight is the camera ray of a new sample: the sample before it is complete when it returns
    PM_FINAL = 0x800u,  // the path of the launch's last sample has ended; only shadow rays are in flight
    PM_VALID = 0x1000u  // the slot holds a pixel
};

struct MegaParams {
    DevScene S;
    WaveBufs W;
    RegionDev R;
    CameraDev cam;
    uint32_t chunk_begin, chunk_n, aa, n_waves;
    int32_t max_depth, shadow_rays;
    uint32_t n_shoot, step_min, trip_limit;
    uint32_t s_begin, s_end;   // the launch runs samples [s_begin, s_end) of every pixel
    float* out;
    const uint32_t* units;     // MEGA_BANKS words per wave: the units the wave owns (MISS: none); unit u = slots [64 u, 64 u + 64)
    const uint32_t* owner;     // per unit: the wave that owns it
    uint32_t* group_cost;      // k_mega_group_cost: cost (a float) per unit
    int32_t scatter_rays;      // != 1: the pixels run pixel_step_tree (one launch, no windows)
    float* stack;              // scatter_rays != 1: max_depth * 20 * W.cap floats, the pixels' path frames
    uint32_t* wave_scale;      // 2 n_waves words: while a wave runs, the clock at its start; then (a float) its time per ray; behind them the waves' times
};
static constexpr uint32_t MEGA_GROUP = 64u, MEGA_BANKS = POOL_PIXELS / MEGA_GROUP;

// Seeds the pixels of a chunk and leaves the camera ray of every pixel's first sample ready (integrator.rs:33-48).
__global__ __launch_bounds__(256) void k_mega_init(MegaParams P) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.chunk_n) return;
    const WaveBufs& W = P.W;
    const uint32_t cap = W.cap;
    uint32_t cx = 0, cy = 0;
    W.p_slot[j] = 0u;
    if (!slot_pixel(P.R, P.chunk_begin + j, cx, cy)) {
        W.q_slot[0][j] = 0u;
        float* o = P.out + 3 * slot_out(P.R, P.chunk_begin + j);
        o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
        return;
    }
    const Rng r = rng_seed((uint64_t)cx ^ __builtin_bswap64((uint64_t)cy));   // integrator.rs:36
    W.rng_lo[j] = r.lo; W.rng_hi[j] = r.hi;
    W.sum[j] = 0.0f; W.sum[j + cap] = 0.0f; W.sum[j + 2 * cap] = 0.0f;
    float* L = W.Lb[0];
    L[j] = 0.0f; L[j + cap] = 0.0f; L[j + 2 * cap] = 0.0f;
    W.beta[j] = 1.0f; W.beta[j + cap] = 1.0f; W.beta[j + 2 * cap] = 1.0f;
    W.p_pix[j] = make_uint2(cx, cy);
    const f3 d = primary_dir(P.cam, cx, cy, 0u, 0u, P.aa);
    float* q = W.q_ray[0];
    q[j] = P.cam.pos.x; q[j + cap] = P.cam.pos.y; q[j + 2 * cap] = P.cam.pos.z;
    q[j + 3 * cap] = d.x; q[j + 4 * cap] = d.y; q[j + 5 * cap] = d.z;
    W.q_slot[0][j] = PM_VALID | PM_RAY;
    W.q_slot[1][j] = 0u;   // (pixel_step_tree: depth 0, waiting for the ray of integrate_recur)
}

// Cost of every unit in the launch before: the rays traced for its pixels since the last call (+ 1 per pixel), priced at the
// time per ray of the wave that owned the unit (rays into the model cost several times those into the sky: the waves' own clocks
// know).  Resets the pixels' ray counts.
__global__ __launch_bounds__(256) void k_mega_group_cost(MegaParams P) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g * MEGA_GROUP >= P.chunk_n) return;
    uint32_t c = 0;
    for (uint32_t k = 0; k < MEGA_GROUP; k++) {
        const uint32_t j = g * MEGA_GROUP + k;
        if (j < P.chunk_n && (P.W.q_slot[0][j] & PM_VALID)) { c += 1u + P.W.p_slot[j]; P.W.p_slot[j] = 0u; }
    }
    const uint32_t w = P.owner[g];
    float scale = w < P.n_waves ? __uint_as_float(P.wave_scale[w]) : 1.0f;
    if (!(scale > 0.0f) || !(scale < 1e30f)) scale = 1.0f;
    P.group_cost[g] = __float_as_uint((float)c * scale);
}

// One step of pixel j's path: everything between the return of the rays it waited for and its next rays.  Called by all
// lanes of the wave from the scheduler's FETCH step; not inlined, so that the scheduler's registers are saved around it
// instead of being shared with it.  Returns the pixel's new pool word (rt_sched.h).
// `params`: the kernel's argument block (the kernel-argument segment, which only a kernel can ask for: it comes here as a
// value); read through the constant address space with wave-uniform addresses, i.e. by scalar loads.
typedef const __attribute__((address_space(4))) uint32_t* const_words;
template <class T>
__device__ __forceinline__ T load_const(const_words src) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    T t;
    uint32_t* d = (uint32_t*)&t;
#pragma unroll
    for (uint32_t k = 0; k < sizeof(T) / 4; k++) d[k] = src[k];
    return t;
}
// (WAVES: the waves per SIMD of the calling kernel.  A function shared by kernels of different register budgets is compiled for the
// tightest of them -- the 80 registers of the mesh-free kernels cost the others a quarter of their rate -- so each budget gets its own.)
template <int WAVES>
__device__ __attribute__((noinline)) uint32_t pixel_step(const uint64_t params, const uint32_t j, const uint32_t active) {
    if (!active) return 0u;
    // (the builtin returns a signed int: without the casts the low half would be sign-extended over the high one)
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(params >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)params);
    const MegaParams PP = load_const<MegaParams>((const_words)pu);
    const MegaParams* P = &PP;
    const DevScene S = P->S;
    const WaveBufs W = P->W;
    const ShadeSet T = W.set[0];
    const uint32_t cap = W.cap;
    const size_t ac = W.any_cap;
    const int32_t shadow_rays = P->shadow_rays;
    if (j >= cap) { W.stats[2] = 3u; return 0u; }   // (cannot happen: the wave's pixels lie inside the arrays)
    uint32_t meta = W.q_slot[0][j];
    uint32_t depth = meta & 0xFFu, sample = meta >> 16;
    float* Lp = W.Lb[0];
    f3 L = mk(Lp[j], Lp[j + cap], Lp[j + 2 * cap]);
    f3 beta = mk(W.beta[j], W.beta[j + cap], W.beta[j + 2 * cap]);
    // everything the step reads whatever the pixel's state is requested here, in one round trip (a step is a chain of dependent
    // loads, and the whole wave waits for it)
    const float* r = W.q_ray[0];
    const f3 pos = mk(r[j], r[j + cap], r[j + 2 * cap]);
    const f3 dir = mk(r[j + 3 * cap], r[j + 4 * cap], r[j + 5 * cap]);
    const uint4 rec0 = W.h_rec[2 * (size_t)j];
    const float rec1 = __uint_as_float(W.h_rec[2 * (size_t)j + 1].x);
    Rng rng;
    rng.lo = W.rng_lo[j]; rng.hi = W.rng_hi[j];
    const f3 sum_in = mk(W.sum[j], W.sum[j + cap], W.sum[j + 2 * cap]);
    const DevObject ob_hit = S.objects[rec0.x < S.n_objects ? rec0.x : 0u];
    asm volatile("" ::: "memory");   // (the loads stay up here)
    // ---- the lit hit whose shadow rays have returned (illumination(), integrator.rs:107-132; :87) ----
    if (meta & PM_LIT) {
        const float* f = T.lit_f;
        const f3 lbeta = mk(f[j], f[j + cap], f[j + 2 * cap]);
        const f3 diffuse = mk(f[j + 3 * cap], f[j + 4 * cap], f[j + 5 * cap]);
        f3 illum = fill(0.0f);
        uint32_t s = 0;
        for (uint32_t l = 0; l < S.n_lights; l++) {
            const DevLight lt = S.lights[l];
            const int ns = light_is_delta(lt) ? 1 : shadow_rays;
            f3 total = fill(0.0f);
            for (int k = 0; k < ns; k++, s++) {
                const float* q = T.smp + (size_t)s * 4 * cap + j;
                const f3 E = mk(lt.emission[0], lt.emission[1], lt.emission[2]);
                float tm;
                f3 ill;
                if (lt.kind == LIGHT_POINT) { tm = q[0]; ill = mk(q[cap], q[2 * (size_t)cap], q[3 * (size_t)cap]); }
                else if (lt.kind == LIGHT_DIRECTION) { tm = 1e12f; ill = E * q[0]; }          // light.rs:120-122
                else if (lt.kind == LIGHT_AO) { tm = lt.param; ill = E; }                    // light.rs:157
                else { tm = 0.0f; ill = E; }                                                 // light.rs:24-27
                const bool visible = (tm == 0.0f) || (T.vis[(size_t)s * cap + j] == 0);      // integrator.rs:124
                if (visible) total = total + ill;
            }
            illum = illum + total / (float)ns;             // integrator.rs:131, :84
        }
        L = L + lbeta * (diffuse * illum);
    }
    // The pixel's part in this launch is over: the image gets the pixel, or, when a launch for later samples follows, the
    // camera ray of its next sample is left ready for that launch (as k_mega_init leaves the first).
    auto leave = [&](const f3 total, const uint32_t sample_done) -> uint32_t {
        const uint32_t aa = P->aa, n_samples = aa * aa;
        if (sample_done + 1u >= n_samples) {
            const float n = (float)n_samples;
            float* o = P->out + 3 * slot_out(P->R, P->chunk_begin + j);
            o[0] = total.x / n; o[1] = total.y / n; o[2] = total.z / n;
            return 0u;
        }
        W.sum[j] = total.x; W.sum[j + cap] = total.y; W.sum[j + 2 * cap] = total.z;
        Lp[j] = 0.0f; Lp[j + cap] = 0.0f; Lp[j + 2 * cap] = 0.0f;
        W.beta[j] = 1.0f; W.beta[j + cap] = 1.0f; W.beta[j + 2 * cap] = 1.0f;
        const uint2 px = W.p_pix[j];
        const CameraDev cam = P->cam;
        const uint32_t sn = sample_done + 1u;
        const f3 d = primary_dir(cam, px.x, px.y, sn / aa, sn % aa, aa);
        float* q = W.q_ray[0];
        q[j] = cam.pos.x; q[j + cap] = cam.pos.y; q[j + 2 * cap] = cam.pos.z;
        q[j + 3 * cap] = d.x; q[j + 4 * cap] = d.y; q[j + 5 * cap] = d.z;
        W.q_slot[0][j] = PM_VALID | PM_RAY | (sn << 16);
        return 0u;
    };
    // ---- a finished sample joins the pixel's sum (integrator.rs:50-53) ----
    f3 sum = mk(0, 0, 0);
    if (meta & (PM_NEWS | PM_FINAL)) {
        sum = sum_in + L;
        L = fill(0.0f);
        beta = fill(1.0f);
        if (meta & PM_FINAL) return leave(sum, sample);
        W.sum[j] = sum.x; W.sum[j + cap] = sum.y; W.sum[j + 2 * cap] = sum.z;
    }
    // ---- the hit of the ray that has returned (integrate_recur, integrator.rs:66-102) ----
    bool lit = false, cont = false;
    f3 next_pos = mk(0, 0, 0), next_dir = mk(0, 0, 0);
    Best b;
    b.obj = rec0.x;
    if (b.obj != MISS && b.obj >= S.n_objects) { W.stats[2] = 4u; b.obj = MISS; }   // (cannot happen: a record no trace has written)
    if (b.obj == MISS) {  // integrator.rs:66
        L = L + beta * sky_color(S, dir);
    } else {
        b.prim = rec0.y;
        b.t_local = __uint_as_float(rec0.z); b.u = __uint_as_float(rec0.w); b.v = rec1;
        f3 hp, hn;
        float hu, hv;
        reconstruct_hit(S, b, pos, dir, hp, hn, hu, hv);
        const uint32_t mat = ob_hit.material;
        const f3 nrm = normalize(hn);                                  // integrator.rs:72-76
        const bool outside = dot(nrm, -dir) > 0.0f;
        const f3 p_out = hp + nrm * 0.001f, p_in = hp - nrm * 0.001f;
        const f3 p = outside ? p_out : p_in;
        const DevMaterial m = S.materials[mat];
        const f3 diffuse = material_diffuse(S, m, hu, hv);
        lit = !is_zero(diffuse);                                       // integrator.rs:80
        if (lit) {                                                     // integrator.rs:80-88
            float* f = T.lit_f;
            f[j] = beta.x; f[j + cap] = beta.y; f[j + 2 * cap] = beta.z;
            f[j + 3 * cap] = diffuse.x; f[j + 4 * cap] = diffuse.y; f[j + 5 * cap] = diffuse.z;
            uint32_t s = 0, s_shoot = 0;
            for (uint32_t l = 0; l < S.n_lights; l++) {
                const DevLight lt = S.lights[l];
                const int ns = light_is_delta(lt) ? 1 : shadow_rays;
                const bool shoots = light_shoots(lt);
                for (int k = 0; k < ns; k++, s++) {
                    f3 sd, ill;
                    float tm;
                    light_sample(lt, p, nrm, rng, sd, tm, ill);
                    float* q = T.smp + (size_t)s * 4 * cap + j;
                    if (lt.kind == LIGHT_POINT) { q[0] = tm; q[cap] = ill.x; q[2 * (size_t)cap] = ill.y; q[3 * (size_t)cap] = ill.z; }
                    else if (lt.kind == LIGHT_DIRECTION) q[0] = rust_max(dot(sd, nrm), 0.0f);   // the factor c of light.rs:120-122
                    if (shoots) {
                        const size_t aq = (size_t)s_shoot * cap + j;
                        float* a = T.a_ray;
                        a[aq] = p.x; a[aq + ac] = p.y; a[aq + 2 * ac] = p.z;
                        a[aq + 3 * ac] = sd.x; a[aq + 4 * ac] = sd.y; a[aq + 5 * ac] = sd.z;
                        a[aq + 6 * ac] = tm;
                        T.a_vidx[aq] = s * cap + j;
                        s_shoot++;
                    }
                }
            }
        }
        f3 out, w;
        if (material_scatter(m, nrm, dir, rng, out, w)) {              // integrator.rs:90-101
            const f3 nb = beta * w;
            if ((int32_t)depth + 1 >= P->max_depth) {                  // integrator.rs:59-61 at depth + 1
                L = L + nb * sky_color(S, out);
            } else {
                cont = true;
                next_pos = dot(out, nrm) > 0.0f ? p_out : p_in;
                next_dir = out;
                beta = nb;
            }
        }
        W.rng_lo[j] = rng.lo; W.rng_hi[j] = rng.hi;
    }
    // ---- what the pixel waits for next ----
    uint32_t flags = PM_VALID | (lit ? PM_LIT : 0u);
    bool ray = cont;
    if (cont) {
        depth++;
    } else {
        if (sample + 1u < P->s_end) {   // the camera ray of the next sample
            sample++;
            depth = 0u;
            const uint2 px = W.p_pix[j];
            const CameraDev cam = P->cam;
            next_pos = cam.pos;
            next_dir = primary_dir(cam, px.x, px.y, sample / P->aa, sample % P->aa, P->aa);
            flags |= PM_NEWS;
            ray = true;
        } else if (lit) {
            flags |= PM_FINAL;
        } else {   // nothing in flight: the pixel leaves here and now
            return leave(((meta & PM_NEWS) ? sum : sum_in) + L, sample);
        }
    }
    if (ray) {
        float* q = W.q_ray[0];
        q[j] = next_pos.x; q[j + cap] = next_pos.y; q[j + 2 * cap] = next_pos.z;
        q[j + 3 * cap] = next_dir.x; q[j + 4 * cap] = next_dir.y; q[j + 5 * cap] = next_dir.z;
        flags |= PM_RAY;
    }
    Lp[j] = L.x; Lp[j + cap] = L.y; Lp[j + 2 * cap] = L.z;
    W.beta[j] = beta.x; W.beta[j + cap] = beta.y; W.beta[j + 2 * cap] = beta.z;
    W.q_slot[0][j] = flags | depth | (sample << 16);
    const uint32_t n_shadow = lit ? P->n_shoot : 0u;
    W.p_slot[j] += (ray ? 1u : 0u) + n_shadow;   // (what the ranges of the next launch are cut by)
    const uint32_t rdy = (ray ? 1u : 0u) | (((1u << n_shadow) - 1u) << 1);
    return PW_ALIVE | rdy | (((ray ? 1u : 0u) + n_shadow) << 8) | (sample << PW_SAMPLE_SHIFT);
}

// ------------------------------------------------------------------------------------------------
// scatter_rays != 1: the reference's recursion (integrator.rs:56-104) draws the random numbers of a whole scatter subtree before
// the next scatter ray of the same hit, so a pixel's rays are a sequence; the pixel is a coroutine with an explicit stack of path
// frames that runs until it needs the result of a closest-hit ray.  The shadow rays of a hit do not stop it (illumination(),
// integrator.rs:107-132, draws all its light samples before the scatter direction, and visibility only enters `color = diffuse *
// illum`): they ride with the pixel's next closest-hit ray and the frame's colour is completed when the pixel resumes; only where
// that colour is needed at once does the pixel wait on its shadow rays alone.  Operation order per pixel is that of
// k_pixel_depth_first and of the round pipeline (k_df_step in rt_kernels.hip, which this follows line by line), so all three produce
// the same bits.  Per pixel: W.q_slot[0] as above (sample index in the high half), W.q_slot[1] = depth | state << 8 | (depth of the
// hit whose shadow rays are out + 1) << 16, a suspended return value in W.Lb[0], the pending hit's diffuse colour in W.beta.
// ------------------------------------------------------------------------------------------------
struct TreeFrame {
    static constexpr uint32_t WORDS = 20;
    f3 color, w, n, dir, p_out, p_in;
    uint32_t mat;
    int32_t k;   // scatter rays drawn so far
};
__device__ __forceinline__ void tree_frame_store(float* stack, size_t cap, uint32_t i, int d, const TreeFrame& f) {
    float* q = stack + (size_t)d * TreeFrame::WORDS * cap + i;
    const float v[TreeFrame::WORDS] = {f.color.x, f.color.y, f.color.z, f.w.x, f.w.y, f.w.z, f.n.x, f.n.y, f.n.z, f.dir.x, f.dir.y, f.dir.z,
                                       f.p_out.x, f.p_out.y, f.p_out.z, f.p_in.x, f.p_in.y, f.p_in.z, __uint_as_float(f.mat), __int_as_float(f.k)};
#pragma unroll
    for (uint32_t c = 0; c < TreeFrame::WORDS; c++) q[c * cap] = v[c];
}
__device__ __forceinline__ void tree_frame_load(const float* stack, size_t cap, uint32_t i, int d, TreeFrame& f) {
    const float* q = stack + (size_t)d * TreeFrame::WORDS * cap + i;
    float v[TreeFrame::WORDS];
#pragma unroll
    for (uint32_t c = 0; c < TreeFrame::WORDS; c++) v[c] = q[c * cap];
    f.color = mk(v[0], v[1], v[2]); f.w = mk(v[3], v[4], v[5]); f.n = mk(v[6], v[7], v[8]); f.dir = mk(v[9], v[10], v[11]);
    f.p_out = mk(v[12], v[13], v[14]); f.p_in = mk(v[15], v[16], v[17]);
    f.mat = __float_as_uint(v[18]); f.k = __float_as_int(v[19]);
}
enum : uint32_t { TS_ENTER = 0, TS_SCATTER = 1, TS_FOLD = 2 };

template <int WAVES>
__device__ __attribute__((noinline)) uint32_t pixel_step_tree(const uint64_t params, const uint32_t slot, const uint32_t active) {
    if (!active) return 0u;
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(params >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)params);
    const MegaParams PP = load_const<MegaParams>((const_words)pu);
    const MegaParams* P = &PP;
    const DevScene S = P->S;
    const WaveBufs W = P->W;
    const ShadeSet T = W.set[0];
    const uint32_t cap = W.cap;
    const size_t ac = W.any_cap;
    if (slot >= cap) { W.stats[2] = 3u; return 0u; }
    float* stack = P->stack;
    float* ret_buf = W.Lb[0];
    const int32_t scatter_rays = P->scatter_rays, max_depth = P->max_depth;
    const float n_scatter = (float)scatter_rays;
    uint32_t pass = W.q_slot[0][slot] >> 16;
    const uint32_t m1 = W.q_slot[1][slot];
    int d = (int)(m1 & 0xFFu);
    const uint32_t state = (m1 >> 8) & 0xFFu, pend_in = m1 >> 16;
    uint32_t pend = 0;
    Rng rng;
    rng.lo = W.rng_lo[slot]; rng.hi = W.rng_hi[slot];
    // what most steps read is requested here, in one round trip with the state words (the whole wave waits for a step, and a step
    // is a chain of dependent loads)
    const float* r_in = W.q_ray[0];
    const f3 ray_pos = mk(r_in[slot], r_in[slot + cap], r_in[slot + 2 * cap]);
    const f3 ray_dir = mk(r_in[slot + 3 * cap], r_in[slot + 4 * cap], r_in[slot + 5 * cap]);
    const uint4 rec0 = W.h_rec[2 * (size_t)slot];
    const float rec1 = __uint_as_float(W.h_rec[2 * (size_t)slot + 1].x);
    const f3 df_in = mk(W.beta[slot], W.beta[slot + cap], W.beta[slot + 2 * cap]);
    asm volatile("" ::: "memory");   // (the loads stay up here)
    TreeFrame f;
    f.color = f.w = f.n = f.dir = f.p_out = f.p_in = fill(0.0f); f.mat = 0; f.k = 0;
    f3 ret = fill(0.0f), rpos = fill(0.0f), rdir = mk(0, 0, 1), diffuse = fill(0.0f), p_lit = fill(0.0f);
    bool lit = false, in_scatter = false, in_return = false;
    if (pend_in) {   // the shadow rays of the pixel's last hit are back: illumination(), integrator.rs:107-132, then :87
        const f3 df = df_in;
        f3 illum = fill(0.0f);
        uint32_t s = 0;
        for (uint32_t l = 0; l < S.n_lights; l++) {
            const DevLight lt = S.lights[l];
            const int ns = light_is_delta(lt) ? 1 : P->shadow_rays;
            f3 tot = fill(0.0f);
            for (int k = 0; k < ns; k++, s++) {
                const float* q = T.smp + (size_t)s * 4 * cap + slot;
                const bool visible = (q[0] == 0.0f) || (T.vis[(size_t)s * cap + slot] == 0);   // integrator.rs:124
                if (visible) tot = tot + mk(q[cap], q[2 * (size_t)cap], q[3 * (size_t)cap]);
            }
            illum = illum + tot / (float)ns;
        }
        const f3 c = fill(0.0f) + df * illum;
        float* q = stack + (size_t)(pend_in - 1u) * TreeFrame::WORDS * cap + slot;
        q[0] = c.x; q[cap] = c.y; q[2 * (size_t)cap] = c.z;   // (the frame's colour: its first three words)
    }
    if (state == TS_ENTER) {               // the closest-hit ray of integrate_recur(ray, d) is back, integrator.rs:64
        rpos = ray_pos;
        rdir = ray_dir;
        if (rec0.x != MISS && rec0.x >= S.n_objects) W.stats[2] = 4u;   // (cannot happen: a record no trace has written)
        if (rec0.x == MISS || rec0.x >= S.n_objects) {              // integrator.rs:66
            ret = sky_color(S, rdir);
            in_return = true;
        } else {
            Best b;
            b.obj = rec0.x; b.prim = rec0.y;
            b.t_local = __uint_as_float(rec0.z); b.u = __uint_as_float(rec0.w); b.v = rec1;
            f3 hp, hn;
            float hu, hv;
            reconstruct_hit(S, b, rpos, rdir, hp, hn, hu, hv);
            f.mat = S.objects[b.obj].material;
            f.n = normalize(hn);
            const bool outside = dot(f.n, -rdir) > 0.0f;
            f.p_out = hp + f.n * 0.001f;
            f.p_in = hp - f.n * 0.001f;
            f.dir = rdir;
            f.color = fill(0.0f);
            f.k = 0;
            p_lit = outside ? f.p_out : f.p_in;
            diffuse = material_diffuse(S, S.materials[f.mat], hu, hv);
            lit = !is_zero(diffuse);       // integrator.rs:80
            in_scatter = true;
        }
    } else {
        tree_frame_load(stack, cap, slot, d, f);
        if (state == TS_FOLD) {            // a scatter ray's radiance waited for this frame's colour, integrator.rs:98
            ret = mk(ret_buf[slot], ret_buf[slot + cap], ret_buf[slot + 2 * cap]);
            f.color = f.color + (f.w * ret) / n_scatter;
        }
        in_scatter = true;
    }
    if (lit) {
        pend = (uint32_t)d + 1u;
        W.beta[slot] = diffuse.x; W.beta[slot + cap] = diffuse.y; W.beta[slot + 2 * cap] = diffuse.z;
        uint32_t s = 0, s_shoot = 0;
        for (uint32_t l = 0; l < S.n_lights; l++) {
            const DevLight lt = S.lights[l];
            const int ns = light_is_delta(lt) ? 1 : P->shadow_rays;
            const bool shoots = light_shoots(lt);
            for (int k = 0; k < ns; k++, s++) {
                f3 sd, ill;
                float tm;
                light_sample(lt, p_lit, f.n, rng, sd, tm, ill);
                float* q = T.smp + (size_t)s * 4 * cap + slot;
                q[0] = tm; q[cap] = ill.x; q[2 * (size_t)cap] = ill.y; q[3 * (size_t)cap] = ill.z;
                if (shoots) {   // (a sample with t_max = 0 is queued as a zero-length ray; the resolve takes it as visible either way)
                    const size_t aq = (size_t)s_shoot * cap + slot;
                    float* a = T.a_ray;
                    a[aq] = p_lit.x; a[aq + ac] = p_lit.y; a[aq + 2 * ac] = p_lit.z;
                    a[aq + 3 * ac] = sd.x; a[aq + 4 * ac] = sd.y; a[aq + 5 * ac] = sd.z;
                    a[aq + 6 * ac] = tm;
                    T.a_vidx[aq] = s * cap + slot;
                    s_shoot++;
                } else {
                    T.vis[(size_t)s * cap + slot] = 0;
                }
            }
        }
    }
    // run the pixel until it needs a ray, waits for its shadow rays, or is finished
    bool want_ray = false, done = false, leaving = false;
    uint32_t new_state = TS_ENTER;
    for (;;) {
        if (in_scatter) {                                   // integrator.rs:90-102
            in_scatter = false;
            bool descend = false;
            while (f.k < scatter_rays) {
                f.k++;
                f3 o, w;
                if (material_scatter(S.materials[f.mat], f.n, f.dir, rng, o, w)) {
                    f.w = w;
                    rpos = dot(o, f.n) > 0.0f ? f.p_out : f.p_in;
                    rdir = o;
                    descend = true;
                    break;
                }
            }
            if (descend) {
                tree_frame_store(stack, cap, slot, d, f);
                d++;
                if (d < max_depth) { want_ray = true; new_state = TS_ENTER; break; }
                ret = sky_color(S, rdir);                   // integrator.rs:59-61: the depth limit ends the path without a ray
                in_return = true;
            } else {
                if (pend == (uint32_t)d + 1u) { tree_frame_store(stack, cap, slot, d, f); new_state = TS_SCATTER; break; }
                ret = f.color;
                in_return = true;
            }
        }
        if (in_return) {                                    // the call at depth d returned `ret`
            in_return = false;
            if (d == 0) {                                   // integrator.rs:50-53: next sample of the pixel
                f3 total = mk(W.sum[slot], W.sum[slot + cap], W.sum[slot + 2 * cap]);
                total = total + ret;
                pass++;
                const uint32_t aa = P->aa;
                if (pass < aa * aa) {
                    W.sum[slot] = total.x; W.sum[slot + cap] = total.y; W.sum[slot + 2 * cap] = total.z;
                    const uint2 px = W.p_pix[slot];
                    const CameraDev cam = P->cam;
                    rpos = cam.pos;
                    rdir = primary_dir(cam, px.x, px.y, pass / aa, pass % aa, aa);
                    want_ray = true; new_state = TS_ENTER;
                    leaving = pass >= P->s_end;   // (that ray is the next launch's)
                } else {
                    total = total / (float)(aa * aa);
                    float* o = P->out + 3 * slot_out(P->R, P->chunk_begin + slot);
                    o[0] = total.x; o[1] = total.y; o[2] = total.z;
                    done = true;
                }
                break;
            }
            d--;
            if (pend == (uint32_t)d + 1u) {                 // that frame's colour is still waiting for its shadow rays
                ret_buf[slot] = ret.x; ret_buf[slot + cap] = ret.y; ret_buf[slot + 2 * cap] = ret.z;
                new_state = TS_FOLD;
                break;
            }
            tree_frame_load(stack, cap, slot, d, f);
            f.color = f.color + (f.w * ret) / n_scatter;
            in_scatter = true;
        }
    }
    if (done) return 0u;
    W.q_slot[0][slot] = PM_VALID | (want_ray ? PM_RAY : 0u) | (pass << 16);
    W.q_slot[1][slot] = (uint32_t)d | (new_state << 8) | (pend << 16);
    W.rng_lo[slot] = rng.lo; W.rng_hi[slot] = rng.hi;
    if (want_ray) {
        float* q = W.q_ray[0];
        q[slot] = rpos.x; q[slot + cap] = rpos.y; q[slot + 2 * cap] = rpos.z;
        q[slot + 3 * cap] = rdir.x; q[slot + 4 * cap] = rdir.y; q[slot + 5 * cap] = rdir.z;
    }
    if (leaving) return 0u;   // (d = 0 here: no hit of this pixel has shadow rays out)
    const uint32_t n_shadow = lit ? P->n_shoot : 0u;
    W.p_slot[slot] += (want_ray ? 1u : 0u) + n_shadow;   // (what the units of the next launch are dealt by)
    const uint32_t rdy = (want_ray ? 1u : 0u) | (((1u << n_shadow) - 1u) << 1);
    return PW_ALIVE | rdy | (((want_ray ? 1u : 0u) + n_shadow) << 8) | ((pass & 0xFFFFu) << PW_SAMPLE_SHIFT);
}

template <int WAVES>
struct MegaIO {
    static constexpr bool MEGA = true;
    const float* q_ray;
    const float* a_ray;
    const uint32_t* a_vidx;
    uint4* h_rec;
    uint8_t* vis;
    unsigned long long* stats;
    uint32_t cap, any_cap;
    const lds_u32* unit_tab;   // the wave's MEGA_BANKS units (LDS); pixel p of the wave is slot 64 unit_tab[p / 64] + p % 64
    uint32_t step_min, trip_limit;
    uint64_t params;     // address of the kernel's argument block
    uint32_t* wave_scale;   // this wave's word of MegaParams::wave_scale
    uint32_t count;      // pixels of the wave
    uint32_t n_waves;
    bool tree;           // scatter_rays != 1
    lds_u32* pix;
    __device__ __forceinline__ void stage_addr(uint32_t, const void* (&src)[8]) const {   // (the queue form's; unused)
#pragma unroll
        for (int c = 0; c < 8; c++) src[c] = q_ray;
    }
    __device__ __forceinline__ void ray_addr(uint32_t ref, const void* (&src)[8]) const {
        const uint32_t k = ref & 7u, j = slot(ref >> 3);
        const bool a = k != 0u;
        const size_t stride = a ? any_cap : cap;
        const float* b0 = a ? a_ray + ((size_t)(k - 1u) * cap + j) : q_ray + j;
#pragma unroll
        for (int c = 0; c < 6; c++) src[c] = b0 + c * stride;
        src[6] = a ? b0 + 6 * stride : b0;
        src[7] = a ? (const void*)(a_vidx + ((size_t)(k - 1u) * cap + j)) : (const void*)b0;
    }
    __device__ __forceinline__ void decode(uint32_t ref, const uint32_t (&raw)[8], f3& pos, f3& dir, float& tmax, bool& any, uint32_t& tag) const {
        pos = mk(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]));
        dir = mk(__uint_as_float(raw[3]), __uint_as_float(raw[4]), __uint_as_float(raw[5]));
        any = (ref & 7u) != 0u;
        tmax = any ? __uint_as_float(raw[6]) : 1e12f;
        tag = raw[7];
    }
    __device__ __forceinline__ void store_closest(uint32_t ref, uint32_t, bool found, const Best& b, float) const {
        uint4* rec = h_rec + 2 * (size_t)slot(ref >> 3);
        if (!found) rec[0].x = MISS;
        if (found) {
            rec[0] = make_uint4(b.obj, b.prim, __float_as_uint(b.t_local), __float_as_uint(b.u));
            rec[1].x = __float_as_uint(b.v);
        }
    }
    __device__ __forceinline__ void store_any(uint32_t, uint32_t tag, bool occluded) const { vis[tag] = occluded ? 1 : 0; }
    __device__ __forceinline__ uint32_t slot(uint32_t p) const { return unit_tab[p >> 6] * MEGA_GROUP + (p & 63u); }
    __device__ __forceinline__ uint32_t step(uint32_t p, bool active) const {
        return tree ? pixel_step_tree<WAVES>(params, slot(p), active ? 1u : 0u) : pixel_step<WAVES>(params, slot(p), active ? 1u : 0u);
    }
    __device__ __forceinline__ void fail(uint32_t lane, uint32_t code) const {
        if (lane == 0u) stats[2] = code;
    }
    __device__ __forceinline__ void finish(uint32_t lane, uint32_t n_closest, uint32_t n_any) const {
        if (lane == 0u) {
            // the wave's time per ray (and pixel): what the next launch's ranges are priced with (k_mega_group_cost)
            const uint32_t ticks = (uint32_t)__builtin_amdgcn_s_memtime() - *wave_scale;
            *wave_scale = __float_as_uint((float)ticks / (float)(n_closest + n_any + count + 1u));
            wave_scale[n_waves] = ticks;   // (for the host's RTAMD_FRAME_DEBUG print)
            atomicAdd(&stats[0], (unsigned long long)n_closest);
            atomicAdd(&stats[1], (unsigned long long)n_any);
        }
    }
};

template <int MESHLOC, int TABKIND, bool LEAN>
__global__ __launch_bounds__(TRACE_BLOCK, MESHLOC == MESH_NONE ? NOMESH_WAVES_PER_SIMD : SCHED_WAVES_PER_SIMD) void k_mega(MegaParams P) {
    const DevScene& S = P.S;
    __shared__ uint32_t pool_mem[TRACE_BLOCK / 64][POOL_WORDS > STAGE_WORDS ? POOL_WORDS : STAGE_WORDS];
    __shared__ unsigned long long coop_mem[TRACE_BLOCK / 64][MESHLOC == MESH_NONE ? 1 : COOP_WORDS / 2];   // (the triangle phase's)
    __shared__ unsigned long long ord_stack[TRACE_BLOCK / 64][sched_stack_depth(MESHLOC, TABKIND == TAB_LDS) ? 64 * sched_stack_depth(MESHLOC, TABKIND == TAB_LDS) : 1];
    lds_u32* pool = (lds_u32*)pool_mem[threadIdx.x >> 6];
    lds_u32* coop = (lds_u32*)coop_mem[threadIdx.x >> 6];
    lds_u32* ostack = (lds_u32*)ord_stack[threadIdx.x >> 6];
    LdsMesh lds;
    LdsTab ltab;
    MidTab mtab;
    stage_scene<MESHLOC, TABKIND>(S, lds, ltab, mtab);
    // wave numbering runs over the blocks first (as the scheduler's)
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * gridDim.x + blockIdx.x);
    const uint32_t lane = lane_id();
    // the wave's units; every pixel of them that has samples left in this launch starts with its camera ray ready (k_mega_init, or
    // the launch before: pixel_step's leave)
    __shared__ uint32_t unit_mem[TRACE_BLOCK / 64][MEGA_BANKS < 8 ? 8 : 16];
    lds_u32* unit_tab = (lds_u32*)unit_mem[threadIdx.x >> 6];
    if (lane < MEGA_BANKS) ((volatile lds_u32*)unit_tab)[lane] = P.units[wave_id * MEGA_BANKS + lane];
    uint32_t count = 0;
#pragma unroll
    for (uint32_t q = 0; q < POOL_PER_LANE; q++) {
        const uint32_t p = POOL_PER_LANE * lane + q;
        const uint32_t unit = ((volatile lds_u32*)unit_tab)[p >> 6];
        const uint32_t j = unit * MEGA_GROUP + (p & 63u);
        uint32_t meta = 0u;
        if (unit != MISS && j < P.chunk_n) meta = P.W.q_slot[0][j];
        const bool go = (meta & (PM_VALID | PM_RAY)) == (PM_VALID | PM_RAY) && (meta >> 16) < P.s_end;
        ((volatile lds_u32*)pool)[p] = go ? (PW_ALIVE | 1u | PW_PEND_ONE | ((meta >> 16) << PW_SAMPLE_SHIFT)) : 0u;
        count += go ? 1u : 0u;
    }
    count = __builtin_amdgcn_readlane(wave_scan_add(count), 63);
    MegaIO<(MESHLOC == MESH_NONE ? NOMESH_WAVES_PER_SIMD : SCHED_WAVES_PER_SIMD)> io{P.W.q_ray[0], P.W.set[0].a_ray, P.W.set[0].a_vidx, P.W.h_rec, P.W.set[0].vis, (unsigned long long*)P.W.stats, P.W.cap, P.W.any_cap, unit_tab, P.step_min, P.trip_limit,
              (uint64_t)(const __attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr(), P.wave_scale + wave_id, count, P.n_waves, P.scatter_rays != 1, pool};
    if (lane == 0u) P.wave_scale[wave_id] = (uint32_t)__builtin_amdgcn_s_memtime();
    if (TABKIND == TAB_LDS) trace_scheduler<MESHLOC, LEAN>(S, ltab, lds, io, nullptr, 0u, 64u, pool, coop, ostack);
    else if (TABKIND == TAB_MID) trace_scheduler<MESHLOC, LEAN>(S, mtab, lds, io, nullptr, 0u, 64u, pool, coop, ostack);
    else trace_scheduler<MESHLOC, LEAN>(S, GlobalTab{&S}, lds, io, nullptr, 0u, 64u, pool, coop, ostack);
}

#ifdef RTAMD_DIAG
hipError_t mega_diag_read(unsigned long long* out48, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_diag), 48 * 8);
    if (e != hipSuccess) return e;
    if (reset) {
        unsigned long long z[48] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, 48 * 8);
    }
    return e;
}
#endif

uint32_t mega_banks() { return MEGA_BANKS; }

uint32_t mega_waves(const DevScene& S, const LaunchCfg& cfg) {
    const bool no_mesh = S.n_meshes == 0;
    const int per_cu = no_mesh ? (cfg.sched_blocks_per_cu * NOMESH_WAVES_PER_SIMD + SCHED_WAVES_PER_SIMD - 1) / SCHED_WAVES_PER_SIMD : cfg.sched_blocks_per_cu;
    return (uint32_t)(cfg.num_cus * per_cu) * (TRACE_BLOCK / 64);
}

static MegaParams mega_params(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, const MegaFrame& F, uint32_t chunk_begin, uint32_t chunk_n,
                              uint32_t s_begin, uint32_t s_end) {
    // A wave that does not come to an end gives up after this many scheduler steps.  The headline frame takes ~300 per sample (64 spp:
    // 20 000 outside the mesh loops, which do not count); the limit leaves three orders of magnitude, for deep paths and large scenes.
    const uint64_t per_sample = (uint64_t)1 << 18;
    const uint64_t lim = per_sample * (s_end > s_begin ? s_end - s_begin : 1u) * (uint64_t)(F.scatter_rays > 1 ? 8 : 1);
    const uint32_t trip_limit = lim > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)lim;
    return MegaParams{S, W, F.R, F.cam, chunk_begin, chunk_n, F.aa, mega_waves(S, cfg), F.max_depth, F.shadow_rays, F.n_shoot, F.step_min, trip_limit, s_begin, s_end,
                      F.out, F.units, F.owner, F.group_cost, F.scatter_rays, F.stack, F.wave_scale};
}

bool mega_supported(const DevScene& S, const LaunchCfg& cfg) {
    (void)S;
    return cfg.scheduler != 0;   // (the frame kernel is the scheduler kernel with a pool in place of the queue)
}

hipError_t launch_mega_init(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, const MegaFrame& F, uint32_t chunk_begin, uint32_t chunk_n, hipStream_t st) {
    if (chunk_n == 0u || chunk_n > W.cap) return hipErrorInvalidValue;
    const MegaParams P = mega_params(S, cfg, W, F, chunk_begin, chunk_n, 0u, 0u);
    hipLaunchKernelGGL(k_mega_init, dim3((chunk_n + 255u) / 256u), dim3(256), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_mega_group_cost(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, const MegaFrame& F, uint32_t chunk_n, hipStream_t st) {
    const MegaParams P = mega_params(S, cfg, W, F, 0u, chunk_n, 0u, 0u);
    const uint32_t n_groups = (chunk_n + MEGA_GROUP - 1u) / MEGA_GROUP;
    hipLaunchKernelGGL(k_mega_group_cost, dim3((n_groups + 255u) / 256u), dim3(256), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_mega(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, const MegaFrame& F, uint32_t chunk_begin, uint32_t chunk_n, uint32_t s_begin,
                       uint32_t s_end, hipStream_t st) {
    const uint32_t n_waves = mega_waves(S, cfg);
    if (chunk_n == 0u || chunk_n > W.cap || F.n_shoot > 7u || F.max_depth < 1 || s_begin >= s_end || s_end > 0xFFFFu) return hipErrorInvalidValue;
    if (!mega_supported(S, cfg)) return hipErrorNotSupported;
    const MegaParams P = mega_params(S, cfg, W, F, chunk_begin, chunk_n, s_begin, s_end);
    const uint32_t g = n_waves / (TRACE_BLOCK / 64);
    RTAMD_DISPATCH_MESHLOC(S, cfg, k_mega, g, st, P);
    return hipGetLastError();
}

// Deals the chunk's units to the waves: by cost, the dearest first, each to the wave with the least so far that has a bank free
// (cost: per unit, null = the first deal: unit u to wave u % n_waves, which gives every wave units from all over the chunk).
// units: MEGA_BANKS words per wave (MISS: bank empty); owner: one word per unit.
void mega_assign(const float* cost, uint32_t chunk_n, uint32_t n_waves, uint32_t* units, uint32_t* owner) {
    const uint32_t n_units = (chunk_n + MEGA_GROUP - 1u) / MEGA_GROUP;
    std::fill(units, units + (size_t)n_waves * MEGA_BANKS, MISS);
    std::vector<uint32_t> held(n_waves, 0u);
    if (!cost) {
        for (uint32_t u = 0; u < n_units; u++) {
            const uint32_t w = u % n_waves;
            units[(size_t)w * MEGA_BANKS + held[w]++] = u;
            owner[u] = w;
        }
        return;
    }
    std::vector<uint32_t> order(n_units);
    for (uint32_t u = 0; u < n_units; u++) order[u] = u;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] != cost[b] ? cost[a] > cost[b] : a < b; });
    typedef std::pair<double, uint32_t> Load;   // (cost so far, wave): the least loaded on top
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (uint32_t w = 0; w < n_waves; w++) heap.push(Load(0.0, w));
    for (uint32_t k = 0; k < n_units; k++) {
        const uint32_t u = order[k];
        const Load top = heap.top();
        heap.pop();
        const uint32_t w = top.second;
        units[(size_t)w * MEGA_BANKS + held[w]++] = u;
        owner[u] = w;
        if (held[w] < MEGA_BANKS) heap.push(Load(top.first + (double)cost[u], w));   // (a full wave leaves the heap; the waves' banks hold more units than a chunk has)
    }
}

}  // namespace rtamd
