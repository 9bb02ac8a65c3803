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

namespace rtamd {

// Per-pixel state in HBM, indexed by j = 256 * wave + p, component-major in the arrays of WaveBufs:
//   rng_lo / rng_hi, sum, Lb[0] (radiance of the sample in work), beta, q_ray[0] (the closest-hit ray in flight),
//   h_rec (its result), p_pix (pixel coordinates), p_slot (pixel slot of the chunk), q_slot[0] (the word below),
//   set[0]: lit_f, smp (the hit whose light samples are in flight), a_ray / a_vidx (shadow ray k at k * cap + j), vis.
// State word: bits 0-7 depth of the ray in flight, bits 16-31 the sample it belongs to, and
enum : uint32_t {
    PM_RAY = 0x100u,    // a closest-hit ray is in flight
    PM_LIT = 0x200u,    // a lit hit waits for the results of its shadow rays
    PM_NEWS = 0x400u,   // the ray in flight is the camera ray of a new sample: the sample before it is complete when it returns
    PM_FINAL = 0x800u,  // the last sample's path has ended; only shadow rays are in flight
    PM_VALID = 0x1000u  // the slot holds a pixel
};

struct MegaParams {
    DevScene S;
    WaveBufs W;
    RegionDev R;
    CameraDev cam;
    uint32_t chunk_begin, chunk_n, aa, n_waves;
    int32_t max_depth, shadow_rays;
    uint32_t n_shoot, step_min;
    float* out;
    uint32_t* park;   // POOL_PARK_WORDS * 64 * n_waves words
};

// slot of the chunk -> (wave, pixel of the wave): units of four consecutive slots are dealt to the waves in turn, so that
// every wave gets an even share of every part of the image
__device__ __forceinline__ uint32_t state_index(uint32_t i, uint32_t n_waves) {
    const uint32_t u = i >> 2, w = u % n_waves, pu = u / n_waves;
    return w * POOL_PIXELS + pu * 4u + (i & 3u);
}

// Seeds the pixels of a chunk and leaves the camera ray of every pixel's first sample ready (integrator.rs:33-48).
__global__ __launch_bounds__(256) void k_mega_init(MegaParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_waves * POOL_PIXELS) return;
    if ((i >> 2) / P.n_waves >= POOL_PIXELS / 4u) return;
    const WaveBufs& W = P.W;
    const uint32_t j = state_index(i, P.n_waves), cap = W.cap;
    uint32_t cx = 0, cy = 0;
    const bool valid = i < P.chunk_n && slot_pixel(P.R, P.chunk_begin + i, cx, cy);
    if (!valid) {
        W.q_slot[0][j] = 0u;
        if (i < P.chunk_n) { float* o = P.out + 3 * (size_t)(P.chunk_begin + i); o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f; }
        return;
    }
    const Rng r = rng_seed((uint64_t)cx ^ __builtin_bswap64((uint64_t)cy));   // integrator.rs:36
    W.rng_lo[j] = r.lo; W.rng_hi[j] = r.hi;
    W.sum[j] = 0.0f; W.sum[j + cap] = 0.0f; W.sum[j + 2 * cap] = 0.0f;
    float* L = W.Lb[0];
    L[j] = 0.0f; L[j + cap] = 0.0f; L[j + 2 * cap] = 0.0f;
    W.beta[j] = 1.0f; W.beta[j + cap] = 1.0f; W.beta[j + 2 * cap] = 1.0f;
    W.p_pix[j] = make_uint2(cx, cy);
    W.p_slot[j] = i;
    const f3 d = primary_dir(P.cam, cx, cy, 0u, 0u, P.aa);
    float* q = W.q_ray[0];
    q[j] = P.cam.pos.x; q[j + cap] = P.cam.pos.y; q[j + 2 * cap] = P.cam.pos.z;
    q[j + 3 * cap] = d.x; q[j + 4 * cap] = d.y; q[j + 5 * cap] = d.z;
    W.q_slot[0][j] = PM_VALID | PM_RAY;
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
    // ---- a finished sample joins the pixel's sum (integrator.rs:50-53) ----
    f3 sum = mk(0, 0, 0);
    if (meta & (PM_NEWS | PM_FINAL)) {
        sum = mk(W.sum[j], W.sum[j + cap], W.sum[j + 2 * cap]) + L;
        L = fill(0.0f);
        beta = fill(1.0f);
        if (meta & PM_FINAL) {
            const float n = (float)(P->aa * P->aa);
            float* o = P->out + 3 * (size_t)(P->chunk_begin + W.p_slot[j]);
            o[0] = sum.x / n; o[1] = sum.y / n; o[2] = sum.z / n;
            return 0u;
        }
        W.sum[j] = sum.x; W.sum[j + cap] = sum.y; W.sum[j + 2 * cap] = sum.z;
    }
    // ---- the hit of the ray that has returned (integrate_recur, integrator.rs:66-102) ----
    const float* r = W.q_ray[0];
    const f3 pos = mk(r[j], r[j + cap], r[j + 2 * cap]);
    const f3 dir = mk(r[j + 3 * cap], r[j + 4 * cap], r[j + 5 * cap]);
    bool lit = false, cont = false;
    f3 next_pos = mk(0, 0, 0), next_dir = mk(0, 0, 0);
    Best b;
    const uint4 rec0 = W.h_rec[2 * (size_t)j];
    b.obj = rec0.x;
    if (b.obj != MISS && b.obj >= S.n_objects) { W.stats[2] = 4u; b.obj = MISS; }   // (cannot happen: a record no trace has written)
    if (b.obj == MISS) {  // integrator.rs:66
        L = L + beta * sky_color(S, dir);
    } else {
        b.prim = rec0.y;
        b.t_local = __uint_as_float(rec0.z); b.u = __uint_as_float(rec0.w); b.v = __uint_as_float(W.h_rec[2 * (size_t)j + 1].x);
        f3 hp, hn;
        float hu, hv;
        reconstruct_hit(S, b, pos, dir, hp, hn, hu, hv);
        const uint32_t mat = S.objects[b.obj].material;
        const f3 nrm = normalize(hn);                                  // integrator.rs:72-76
        const bool outside = dot(nrm, -dir) > 0.0f;
        const f3 p_out = hp + nrm * 0.001f, p_in = hp - nrm * 0.001f;
        const f3 p = outside ? p_out : p_in;
        const DevMaterial m = S.materials[mat];
        const f3 diffuse = material_diffuse(S, m, hu, hv);
        lit = !is_zero(diffuse);                                       // integrator.rs:80
        Rng rng;
        rng.lo = W.rng_lo[j]; rng.hi = W.rng_hi[j];
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
        const uint32_t n_samples = P->aa * P->aa;
        if (sample + 1u < n_samples) {   // the camera ray of the next sample
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
        } else {   // nothing in flight: the pixel is finished here and now
            sum = mk(W.sum[j], W.sum[j + cap], W.sum[j + 2 * cap]) + L;
            const float n = (float)n_samples;
            float* o = P->out + 3 * (size_t)(P->chunk_begin + W.p_slot[j]);
            o[0] = sum.x / n; o[1] = sum.y / n; o[2] = sum.z / n;
            return 0u;
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
    const uint32_t rdy = (ray ? 1u : 0u) | (((1u << n_shadow) - 1u) << 1);
    return PW_ALIVE | rdy | (((ray ? 1u : 0u) + n_shadow) << 8) | (sample << PW_SAMPLE_SHIFT);
}

struct MegaIO {
    static constexpr bool MEGA = true;
    const float* q_ray;
    const float* a_ray;
    const uint32_t* a_vidx;
    uint4* h_rec;
    uint8_t* vis;
    unsigned long long* stats;
    uint32_t cap, any_cap;
    uint32_t base;       // state index of the wave's first pixel
    uint32_t step_min;
    uint64_t params;     // address of the kernel's argument block
    uint32_t* park;      // lane state parked around pixel_step (rt_sched.h)
    size_t park_stride;
    lds_u32* pix;
    __device__ __forceinline__ void stage_addr(uint32_t, const void* (&src)[8]) const {   // (the queue form's; unused)
#pragma unroll
        for (int c = 0; c < 8; c++) src[c] = q_ray;
    }
    __device__ __forceinline__ void ray_addr(uint32_t ref, const void* (&src)[8]) const {
        const uint32_t k = ref & 7u, j = base + (ref >> 3);
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
        uint4* rec = h_rec + 2 * (size_t)(base + (ref >> 3));
        if (!found) rec[0].x = MISS;
        if (found) {
            rec[0] = make_uint4(b.obj, b.prim, __float_as_uint(b.t_local), __float_as_uint(b.u));
            rec[1].x = __float_as_uint(b.v);
        }
    }
    __device__ __forceinline__ void store_any(uint32_t, uint32_t tag, bool occluded) const { vis[tag] = occluded ? 1 : 0; }
    __device__ __forceinline__ uint32_t step(uint32_t p, bool active) const { return pixel_step(params, base + p, active ? 1u : 0u); }
    __device__ __forceinline__ void fail(uint32_t lane, uint32_t code) const {
        if (lane == 0u) stats[2] = code;
    }
    __device__ __forceinline__ void finish(uint32_t lane, uint32_t n_closest, uint32_t n_any) const {
        if (lane == 0u) {
            atomicAdd(&stats[0], (unsigned long long)n_closest);
            atomicAdd(&stats[1], (unsigned long long)n_any);
        }
    }
};

template <int MESHLOC, int TABKIND, bool LEAN>
__global__ __launch_bounds__(TRACE_BLOCK, MESHLOC == MESH_NONE ? NOMESH_WAVES_PER_SIMD : SCHED_WAVES_PER_SIMD) void k_mega(MegaParams P) {
    const DevScene& S = P.S;
    __shared__ uint32_t pool_mem[TRACE_BLOCK / 64][STAGE_WORDS];
    __shared__ unsigned long long coop_mem[TRACE_BLOCK / 64][MESHLOC == MESH_NONE ? 1 : COOP_WORDS / 2];   // (the triangle phase's)
    __shared__ unsigned long long ord_stack[TRACE_BLOCK / 64][sched_stack_depth(MESHLOC, TABKIND == TAB_LDS) ? 64 * sched_stack_depth(MESHLOC, TABKIND == TAB_LDS) : 1];
    static_assert(POOL_WORDS <= STAGE_WORDS, "the pool takes the place of the ray staging buffer");
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
    const uint32_t base = wave_id * POOL_PIXELS;
    // every pixel of the wave starts with its first camera ray ready (k_mega_init)
#pragma unroll
    for (uint32_t q = 0; q < 4u; q++) {
        const uint32_t p = 4u * lane + q;
        const uint32_t meta = P.W.q_slot[0][base + p];
        ((volatile lds_u32*)pool)[p] = (meta & PM_VALID) ? (PW_ALIVE | 1u | PW_PEND_ONE) : 0u;
    }
    MegaIO io{P.W.q_ray[0], P.W.set[0].a_ray, P.W.set[0].a_vidx, P.W.h_rec, P.W.set[0].vis, (unsigned long long*)P.W.stats, P.W.cap, P.W.any_cap, base, P.step_min,
              (uint64_t)(const __attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr(), P.park, (size_t)P.n_waves * 64u, pool};
    if (TABKIND == TAB_LDS) trace_scheduler<MESHLOC, LEAN>(S, ltab, lds, io, nullptr, 0u, 64u, pool, coop, ostack);
    else if (TABKIND == TAB_MID) trace_scheduler<MESHLOC, LEAN>(S, mtab, lds, io, nullptr, 0u, 64u, pool, coop, ostack);
    else trace_scheduler<MESHLOC, LEAN>(S, GlobalTab{&S}, lds, io, nullptr, 0u, 64u, pool, coop, ostack);
}

uint32_t mega_waves(const DevScene& S, const LaunchCfg& cfg) {
    const bool no_mesh = S.n_meshes == 0;
    const int per_cu = no_mesh ? (cfg.sched_blocks_per_cu * NOMESH_WAVES_PER_SIMD + SCHED_WAVES_PER_SIMD - 1) / SCHED_WAVES_PER_SIMD : cfg.sched_blocks_per_cu;
    return (uint32_t)(cfg.num_cus * per_cu) * (TRACE_BLOCK / 64);
}

hipError_t launch_mega(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, const MegaFrame& F, uint32_t chunk_begin, uint32_t chunk_n, hipStream_t st) {
    const uint32_t n_waves = mega_waves(S, cfg);
    if (chunk_n == 0u || chunk_n > n_waves * POOL_PIXELS || F.n_shoot > 7u || F.max_depth < 1) return hipErrorInvalidValue;
    MegaParams P{S, W, F.R, F.cam, chunk_begin, chunk_n, F.aa, n_waves, F.max_depth, F.shadow_rays, F.n_shoot, F.step_min, F.out, F.park};
    const uint32_t n_state = n_waves * POOL_PIXELS;
    const uint32_t g = n_waves / (TRACE_BLOCK / 64);
#ifdef RTAMD_MEGA_ALL
    hipLaunchKernelGGL(k_mega_init, dim3((n_state + 255u) / 256u), dim3(256), 0, st, P);
    RTAMD_DISPATCH_MESHLOC(S, cfg, k_mega, g, st, P);
#else
    {   // (development build: the bunny scene's instance only)
        const int loc_ = mesh_location(S, cfg);
        const int tab_ = !cfg.use_lds ? TAB_GLOBAL : (S.tab_bytes > 0 ? TAB_LDS : (S.mid_bytes > 0 ? TAB_MID : TAB_GLOBAL));
        if (loc_ != MESH_LDS || tab_ != TAB_LDS) return hipErrorNotSupported;
        const uint32_t lb_ = S.lds_sched_bytes + S.tab_bytes;
        const bool lean_ = cfg.lean && (S.features & ~sched_features(loc_, true)) == 0u;
        hipLaunchKernelGGL(k_mega_init, dim3((n_state + 255u) / 256u), dim3(256), 0, st, P);
        RTAMD_DISPATCH_LEAN(k_mega, MESH_LDS, TAB_LDS, lean_, g, lb_, st, P);
    }
#endif
    return hipGetLastError();
}

}  // namespace rtamd
