// HIP kernels of the render hot path for MI355X (gfx950): batched closest-hit / any-hit tracing and
// the wavefront form of the reference integrator (ray generation, trace, shade + light sampling +
// skybox lookup, shadow trace, wave64 ballot compaction of the surviving paths).
//
// Work distribution: one lane = one ray / path.  Every queue-driven kernel is launched with a fixed
// grid that covers the chip and walks its queue with a grid stride; the queue length is read from
// device memory, so the host never synchronises inside a frame.
#include "rt_kernels.h"
#include <algorithm>
#include <vector>
#include "rt_sched.h"
#include "rt_stage.h"
#include "rt_integrate.h"

namespace rtamd {

// ------------------------------------------------------------------------------------------------
// LDS staging of one small mesh (nodes + triangles): every block copies it once, then all its
// waves traverse out of LDS instead of L1/L2.  ds_read_b128 per node half / triangle third.
// ------------------------------------------------------------------------------------------------
struct LdsMem {
    const float4* nodes;
    const float4* tris;
    __device__ __forceinline__ void node(uint32_t i, float4& n0, float4& n1) const { n0 = nodes[2 * i]; n1 = nodes[2 * i + 1]; }
    __device__ __forceinline__ void tri(uint32_t k, float4& q0, float4& q1, float4& q2) const {
        q0 = tris[3 * k]; q1 = tris[3 * k + 1]; q2 = tris[3 * k + 2];
    }
};

template <bool USE_LDS>
__device__ __forceinline__ const LdsMem* stage_mesh(const DevScene& S, LdsMem& mem) {
    if (!USE_LDS) return nullptr;
    const DevMesh m = S.meshes[S.lds_mesh];
    const uint32_t n_node4 = 2 * m.n_nodes;
    const uint32_t n_tri4 = S.lds_bytes / 16 - n_node4;
    const float4* gn = S.nodes + 2 * (size_t)m.node_base;
    const float4* gt = S.tris + 3 * (size_t)m.tri_base;
    for (uint32_t i = threadIdx.x; i < n_node4; i += blockDim.x) g_lds[i] = gn[i];
    for (uint32_t i = threadIdx.x; i < n_tri4; i += blockDim.x) g_lds[n_node4 + i] = gt[i];
    __syncthreads();
    mem.nodes = g_lds;
    mem.tris = g_lds + n_node4;
    return &mem;
}

// ------------------------------------------------------------------------------------------------
// Batched Geometry::hit / is_hit on caller-provided ray arrays (the parity entry points).
// ------------------------------------------------------------------------------------------------
template <bool USE_LDS>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_closest(DevScene S, RayBatch rays, HitBatch out, uint32_t n) {
    LdsMem mem;
    const LdsMem* lm = stage_mesh<USE_LDS>(S, mem);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        f3 pos = mk(rays.ox[i], rays.oy[i], rays.oz[i]);
        f3 dir = mk(rays.dx[i], rays.dy[i], rays.dz[i]);
        Best b;
        b.obj = MISS; b.prim = 0; b.t_local = 0.f; b.u = 0.f; b.v = 0.f;
        float t_hit;
        bool found = scene_closest(S, lm, pos, dir, rays.tmax[i], b, t_hit);
        float hu = 0.f, hv = 0.f;
        if (found) {
            f3 hp, hn;
            reconstruct_hit(S, b, pos, dir, hp, hn, hu, hv);
        }
        out.obj[i] = found ? b.obj : MISS;
        out.prim[i] = found ? b.prim : MISS;
        out.t[i] = found ? t_hit : 0.0f;
        out.u[i] = hu;
        out.v[i] = hv;
    }
}

template <bool USE_LDS>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_any(DevScene S, RayBatch rays, uint8_t* out, uint32_t n) {
    LdsMem mem;
    const LdsMem* lm = stage_mesh<USE_LDS>(S, mem);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        f3 pos = mk(rays.ox[i], rays.oy[i], rays.oz[i]);
        f3 dir = mk(rays.dx[i], rays.dy[i], rays.dz[i]);
        out[i] = scene_any(S, lm, pos, dir, rays.tmax[i]) ? 1 : 0;
    }
}

// Scheduler versions of the two batch kernels (see rt_sched.h).

struct BatchIO {
    static constexpr bool MEGA = false;
    DevScene S;
    RayBatch rays;
    HitBatch hits;
    uint8_t* any_out;
    bool any_mode;
    __device__ __forceinline__ void stage_addr(uint32_t i, const void* (&src)[8]) const {
        src[0] = rays.ox + i; src[1] = rays.oy + i; src[2] = rays.oz + i;
        src[3] = rays.dx + i; src[4] = rays.dy + i; src[5] = rays.dz + i;
        src[6] = rays.tmax + i; src[7] = rays.tmax + i;
    }
    __device__ __forceinline__ void decode(uint32_t, const uint32_t (&raw)[8], f3& pos, f3& dir, float& tmax, bool& any, uint32_t& tag) const {
        pos = mk(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]));
        dir = mk(__uint_as_float(raw[3]), __uint_as_float(raw[4]), __uint_as_float(raw[5]));
        tmax = __uint_as_float(raw[6]);
        any = any_mode;
        tag = 0;
    }
    __device__ __forceinline__ void store_closest(uint32_t i, uint32_t, bool found, const Best& b, float t_world) const {
        float hu = b.u, hv = b.v;
        if (found && S.objects[b.obj].geom_kind != GEOM_MESH && S.objects[b.obj].geom_kind != GEOM_TRIANGLE) {
            f3 hp, hn;  // sphere / cuboid uv need the hit point
            reconstruct_hit(S, b, mk(rays.ox[i], rays.oy[i], rays.oz[i]), mk(rays.dx[i], rays.dy[i], rays.dz[i]), hp, hn, hu, hv);
        }
        hits.obj[i] = found ? b.obj : MISS;
        hits.prim[i] = found ? b.prim : MISS;
        hits.t[i] = found ? t_world : 0.0f;
        hits.u[i] = found ? hu : 0.0f;
        hits.v[i] = found ? hv : 0.0f;
    }
    __device__ __forceinline__ void store_any(uint32_t i, uint32_t, bool occluded) const { any_out[i] = occluded ? 1 : 0; }
};

template <int MESHLOC, int TABKIND, bool LEAN>
__global__ __launch_bounds__(TRACE_BLOCK, MESHLOC == MESH_NONE ? NOMESH_WAVES_PER_SIMD : SCHED_WAVES_PER_SIMD) void k_trace_batch_sched(DevScene S, BatchIO io, uint32_t n, uint32_t* work_head, uint32_t chunk) {
    {   // not even the block's first wave has a batch (see trace_scheduler)
        const uint32_t bs = sched_batch_size(n, gridDim.x * (TRACE_BLOCK / 64), tune_min_batch(chunk));
        if (blockIdx.x >= (n + bs - 1u) / bs) return;
    }
    __shared__ uint32_t ray_stage[TRACE_BLOCK / 64][STAGE_WORDS];
    __shared__ unsigned long long coop_mem[TRACE_BLOCK / 64][MESHLOC == MESH_NONE ? 1 : COOP_WORDS / 2];   // (the triangle phase's)
    __shared__ unsigned long long ord_stack[TRACE_BLOCK / 64][sched_stack_depth(MESHLOC, TABKIND == TAB_LDS) ? 64 * sched_stack_depth(MESHLOC, TABKIND == TAB_LDS) : 1];
    LdsMesh lds;
    LdsTab ltab;
    MidTab mtab;
    stage_scene<MESHLOC, TABKIND>(S, lds, ltab, mtab);
    lds_u32* stage = (lds_u32*)ray_stage[threadIdx.x >> 6];
    lds_u32* coop = (lds_u32*)coop_mem[threadIdx.x >> 6];
    lds_u32* ostack = (lds_u32*)ord_stack[threadIdx.x >> 6];
    if (TABKIND == TAB_LDS) trace_scheduler<MESHLOC, LEAN>(S, ltab, lds, io, work_head, n, chunk, stage, coop, ostack);
    else if (TABKIND == TAB_MID) trace_scheduler<MESHLOC, LEAN>(S, mtab, lds, io, work_head, n, chunk, stage, coop, ostack);
    else trace_scheduler<MESHLOC, LEAN>(S, GlobalTab{&S}, lds, io, work_head, n, chunk, stage, coop, ostack);
}


// Diagnostic build only: time stamps (s_memtime, 100 MHz) along the shade kernel's dependency chain, taken by the
// first thread of the first block; slot k accumulates the ticks between probe k-1 and probe k, slot 16 + k counts.
#ifdef RTAMD_DIAG
__device__ unsigned long long g_shade_diag[32];
#define SHADE_PROBE_DECL unsigned long long sp_t = __builtin_amdgcn_s_memtime(); const bool sp_on = blockIdx.x == 0 && threadIdx.x == 0;
#define SHADE_PROBE(k) { unsigned long long sp_n = __builtin_amdgcn_s_memtime(); if (sp_on) { g_shade_diag[k] += sp_n - sp_t; g_shade_diag[16 + k] += 1; } sp_t = sp_n; }
#else
#define SHADE_PROBE_DECL
#define SHADE_PROBE(k)
#endif

// Block-wide compaction: threads with `take` get consecutive slots of a global queue.  The waves
// of a block pool their counts in LDS and the block issues ONE atomic.  A single queue counter sustains only ~90 atomics/us, so stages that
// append from every wave are bound by it unless they aggregate.  Every thread of the block must
// call this (it synchronises); `sh` holds one word per wave.
template <int NWAVES>
__device__ __forceinline__ uint32_t block_append(bool take, uint32_t* counter, uint32_t* sh) {
    static_assert(NWAVES <= 64, "one scan lane per wave");
    unsigned long long mask = __ballot(take);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t before = __popcll(mask & ((1ULL << lane) - 1ULL));
    if (lane == 0) sh[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (wave == 0) {
        const uint32_t v = lane < (uint32_t)NWAVES ? sh[lane] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < NWAVES; d <<= 1) {
            uint32_t t = __shfl_up(incl, d);
            if ((int)lane >= d) incl += t;
        }
        const uint32_t sum = __shfl(incl, NWAVES - 1);
        uint32_t base = 0;
        if (lane == 0 && sum) base = atomicAdd(counter, sum);
        base = __shfl(base, 0);
        if (lane < (uint32_t)NWAVES) sh[lane] = base + incl - v;
    }
    __syncthreads();
    // sh[wave] is next overwritten by this same wave, so no third barrier is needed
    return sh[wave] + before;
}

// As block_append, and also hands out the range the whole block got (`base`, `count`), so that a second queue whose
// entries are a fixed multiple `mult` of this one's needs no counter traffic of its own beyond one fire-and-forget
// add of count * mult to `counter2`.  `sh` holds two words more than one per wave.
template <int NWAVES>
__device__ __forceinline__ uint32_t block_append_range(bool take, uint32_t* counter, uint32_t* counter2, uint32_t mult, uint32_t* sh, uint32_t& base_out,
                                                       uint32_t& count_out) {
    static_assert(NWAVES <= 62, "one scan lane per wave");
    unsigned long long mask = __ballot(take);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t before = __popcll(mask & ((1ULL << lane) - 1ULL));
    if (lane == 0) sh[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (wave == 0) {
        const uint32_t v = lane < (uint32_t)NWAVES ? sh[lane] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < NWAVES; d <<= 1) {
            uint32_t t = __shfl_up(incl, d);
            if ((int)lane >= d) incl += t;
        }
        const uint32_t sum = __shfl(incl, NWAVES - 1);
        uint32_t base = 0;
        if (lane == 0 && sum) {
            base = atomicAdd(counter, sum);
            if (mult) atomicAdd(counter2, sum * mult);   // result unused: no wait
        }
        base = __shfl(base, 0);
        if (lane < (uint32_t)NWAVES) sh[lane] = base + incl - v;
        if (lane == 0) { sh[NWAVES] = base; sh[NWAVES + 1] = sum; }
    }
    __syncthreads();
    base_out = sh[NWAVES];
    count_out = sh[NWAVES + 1];
    return sh[wave] + before;
}


// ------------------------------------------------------------------------------------------------
// Stage 1: ray generation.  One lane per pixel slot of the chunk; sample (si, sj) of the regular
// antialiasing grid (src/integrator.rs:33-48).  Primary rays depend on nothing but the pixel and the
// sample index, so this stage only fills the primary queue; the per-sample reset of the pixel's
// radiance and throughput is done by the shade launch of bounce 0 (the primary rays of a pass are
// generated and traced while the pass before it is still in flight).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(STAGE_BLOCK) void k_raygen(WaveBufs W, RegionDev R, CameraDev cam, uint32_t chunk_begin, uint32_t chunk_n,
                                                uint32_t si, uint32_t sj, uint32_t aa, uint32_t first, uint32_t p_idx) {
    __shared__ uint32_t sh_app[STAGE_BLOCK / 64];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = false;
    uint32_t cx = 0, cy = 0;
    if (i < chunk_n) valid = slot_pixel(R, chunk_begin + i, cx, cy);
    f3 pos = mk(0, 0, 0), dir = mk(0, 0, 0);
    if (valid) {
        if (first) {
            uint64_t seed = (uint64_t)cx ^ __builtin_bswap64((uint64_t)cy);  // integrator.rs:36
            Rng r = rng_seed(seed);
            W.rng_lo[i] = r.lo; W.rng_hi[i] = r.hi;
            W.sum[i] = 0.0f; W.sum[i + W.cap] = 0.0f; W.sum[i + 2 * W.cap] = 0.0f;
        }
        dir = primary_dir(cam, cx, cy, si, sj, aa);
        pos = cam.pos;
    }
    uint32_t q = block_append<STAGE_BLOCK / 64>(valid, &W.counts[p_idx], sh_app);
    if (valid) {
        W.p_slot[q] = i;
        W.p_pix[q] = make_uint2(cx, cy);
        float* r = W.p_ray;
        r[q] = pos.x; r[q + W.cap] = pos.y; r[q + 2 * W.cap] = pos.z;
        r[q + 3 * W.cap] = dir.x; r[q + 4 * W.cap] = dir.y; r[q + 5 * W.cap] = dir.z;
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 2: trace.  One persistent launch serves up to four queues: the closest-hit rays of a bounce
// (scene.root.hit(ray, 1e12), integrator.rs:64), the primary rays of the next sample pass, and the
// shadow rays (scene.root.is_hit(&ray, t_max), integrator.rs:124) that the shade launches before it
// left in the two shade sets.
// ------------------------------------------------------------------------------------------------
struct WaveIO {
    static constexpr bool MEGA = false;
    const float* q_ray;   // closest-hit ray queue of this bounce -> h_rec
    const float* p_ray;   // primary rays -> p_rec
    const float *a_ray0, *a_ray1;   // shadow-ray queues of the two shade sets (separate members: an array would be indexed through scratch memory)
    const uint32_t *a_vidx0, *a_vidx1;
    uint4* h_rec;
    uint4* p_rec;
    uint8_t *vis0, *vis1;
    uint32_t cap, any_cap;
    // Items [0, n_q) are bounce rays, [n_q, n_closest) primary rays, [n_closest, n_a0) shadow rays of the first shade set, the rest
    // shadow rays of the second (words 6 and 7 of a closest-hit ray are ignored and point at its first word).
    uint32_t n_q, n_closest, n_a0;
    __device__ __forceinline__ void stage_addr(uint32_t i, const void* (&src)[8]) const {
        const bool a = i >= n_closest;
        const bool second = i >= (a ? +n_a0 : +n_q);
        const uint32_t j = i - (a ? (second ? +n_a0 : +n_closest) : (second ? +n_q : 0u));
        // unary plus: a conditional over two members is an lvalue, and the nested choice of addresses would go through a
        // table in scratch memory
        const float* ra = second ? +a_ray1 : +a_ray0;
        const float* rc = second ? +p_ray : +q_ray;
        const float* base = (a ? ra : rc) + j;
        const size_t stride = a ? any_cap : cap;
#pragma unroll
        for (int c = 0; c < 6; c++) src[c] = base + c * stride;
        src[6] = a ? base + 6 * stride : base;
        src[7] = a ? (const void*)((second ? +a_vidx1 : +a_vidx0) + j) : (const void*)base;
    }
    __device__ __forceinline__ void decode(uint32_t i, const uint32_t (&raw)[8], f3& pos, f3& dir, float& tmax, bool& any, uint32_t& tag) const {
        pos = mk(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]));
        dir = mk(__uint_as_float(raw[3]), __uint_as_float(raw[4]), __uint_as_float(raw[5]));
        any = i >= n_closest;
        tmax = any ? __uint_as_float(raw[6]) : 1e12f;
        tag = raw[7];
    }
    __device__ __forceinline__ void store_closest(uint32_t i, uint32_t, bool found, const Best& b, float) const {
        uint4* rec = i >= n_q ? +p_rec + 2 * (size_t)(i - n_q) : +h_rec + 2 * (size_t)i;
        if (!found) rec[0].x = MISS;
        if (found) {
            rec[0] = make_uint4(b.obj, b.prim, __float_as_uint(b.t_local), __float_as_uint(b.u));
            rec[1].x = __float_as_uint(b.v);
        }
    }
    __device__ __forceinline__ void store_any(uint32_t i, uint32_t tag, bool occluded) const { (i >= n_a0 ? +vis1 : +vis0)[tag] = occluded ? 1 : 0; }
};

template <int MESHLOC, int TABKIND, bool LEAN>
__global__ __launch_bounds__(TRACE_BLOCK, MESHLOC == MESH_NONE ? NOMESH_WAVES_PER_SIMD : SCHED_WAVES_PER_SIMD) void k_wave_trace(DevScene S, WaveBufs W, ShadeSet A0, ShadeSet A1, uint32_t cur, uint32_t q_idx, uint32_t p_idx, uint32_t a0_idx,
                                                            uint32_t a1_idx, uint32_t head_idx, uint32_t chunk_hint) {
    // queue lengths written by the previous stages (index MISS: that queue is absent)
    const uint32_t n_q = q_idx == MISS ? 0u : W.counts[q_idx];
    const uint32_t n_p = p_idx == MISS ? 0u : W.counts[p_idx];
    const uint32_t n_a0 = a0_idx == MISS ? 0u : W.counts[a0_idx];
    const uint32_t n_a1 = a1_idx == MISS ? 0u : W.counts[a1_idx];
    const uint32_t n_closest = n_q + n_p;
    const uint32_t n = n_closest + n_a0 + n_a1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        W.stats[0] += n_closest;
        W.stats[1] += n_a0 + n_a1;
    }
    {   // not even the block's first wave has a batch (see trace_scheduler)
        const uint32_t bs = sched_batch_size(n, gridDim.x * (TRACE_BLOCK / 64), tune_min_batch(chunk_hint));
        if (n == 0 || blockIdx.x >= (n + bs - 1u) / bs) return;
    }
    const uint32_t chunk = chunk_hint;
    __shared__ uint32_t ray_stage[TRACE_BLOCK / 64][STAGE_WORDS];
    __shared__ unsigned long long coop_mem[TRACE_BLOCK / 64][MESHLOC == MESH_NONE ? 1 : COOP_WORDS / 2];   // (the triangle phase's)
    __shared__ unsigned long long ord_stack[TRACE_BLOCK / 64][sched_stack_depth(MESHLOC, TABKIND == TAB_LDS) ? 64 * sched_stack_depth(MESHLOC, TABKIND == TAB_LDS) : 1];
    lds_u32* stage = (lds_u32*)ray_stage[threadIdx.x >> 6];
    lds_u32* coop = (lds_u32*)coop_mem[threadIdx.x >> 6];
    lds_u32* ostack = (lds_u32*)ord_stack[threadIdx.x >> 6];
    LdsMesh lds;
    LdsTab ltab;
    MidTab mtab;
    stage_scene<MESHLOC, TABKIND>(S, lds, ltab, mtab);
    WaveIO io{cur ? W.q_ray[1] : W.q_ray[0], W.p_ray, A0.a_ray, A1.a_ray, A0.a_vidx, A1.a_vidx, W.h_rec, W.p_rec,
              A0.vis, A1.vis, W.cap, W.any_cap, n_q, n_closest, n_closest + n_a0};
    if (TABKIND == TAB_LDS) trace_scheduler<MESHLOC, LEAN>(S, ltab, lds, io, &W.counts[head_idx], n, chunk, stage, coop, ostack);
    else if (TABKIND == TAB_MID) trace_scheduler<MESHLOC, LEAN>(S, mtab, lds, io, &W.counts[head_idx], n, chunk, stage, coop, ostack);
    else trace_scheduler<MESHLOC, LEAN>(S, GlobalTab{&S}, lds, io, &W.counts[head_idx], n, chunk, stage, coop, ostack);
}

// ------------------------------------------------------------------------------------------------
// Stage 3: shade (integrate_recur after the hit, src/integrator.rs:66-102 with scatter_rays = 1).
// Miss -> skybox, path ends.  Hit -> rebuild the HitResult, diffuse colour; if non-black, draw the
// light samples in light order (their shadow rays are appended to the any-hit queue one by one),
// then draw the scatter direction and append the continuation ray to the next path queue.  The
// per-pixel random stream is consumed in exactly the order of the reference's recursion.
// ------------------------------------------------------------------------------------------------
// `T` is the shade set the launch fills.  Bounce 0 (`fold` != 0) reads the primary queue and starts the pixel's next
// sample: fold 2 adds the radiance buffer's old content (the sample two passes back, complete by now) to the
// pixel sum in generation order (integrator.rs:50-53), fold 1 and 2 zero the buffer; the throughput starts at 1.
// (block `bid` of `n_blocks` blocks of the launch that shade)
__device__ __forceinline__ void shade_part(const DevScene& S, const WaveBufs& W, const ShadeSet& T, int32_t depth, int32_t max_depth, int32_t shadow_rays,
                                           uint32_t fold, uint32_t q_in_idx, uint32_t q_out_idx, uint32_t lit_idx, uint32_t any_idx, uint32_t n_shoot,
                                           const NextPass& next, const uint32_t bid, const uint32_t n_blocks) {
    __shared__ uint32_t sh_app[STAGE_BLOCK / 64 + 2];
    SHADE_PROBE_DECL
    const uint32_t n = W.counts[q_in_idx];
    if (next.zero_base != MISS && bid == 0)   // nothing reads or appends to that set before this launch is over
        for (uint32_t k = threadIdx.x; k < CounterSlots::TOTAL; k += blockDim.x) W.counts[next.zero_base + k] = 0u;
    const bool odd = (depth & 1) != 0;
    const float* r = fold ? W.p_ray : (odd ? W.q_ray[1] : W.q_ray[0]);
    const uint32_t* q_slot_in = fold ? W.p_slot : (odd ? W.q_slot[1] : W.q_slot[0]);
    const uint4* rec_in = fold ? W.p_rec : W.h_rec;
    float* q_ray_out = odd ? W.q_ray[0] : W.q_ray[1];
    uint32_t* q_slot_out = odd ? W.q_slot[0] : W.q_slot[1];
    const uint32_t cap = W.cap;
    const size_t ac = W.any_cap;
    const uint32_t stride = n_blocks * blockDim.x;
    const uint32_t n_iter = (n + stride - 1) / stride;
    for (uint32_t it = 0; it < n_iter; it++) {
        // ---- phase A: load the path, classify (miss / lit / unlit) ----
        uint32_t i = it * stride + bid * blockDim.x + threadIdx.x;
        bool active = i < n;
        bool hit = false, lit = false, cont = false;
        uint32_t slot = 0, mat = MISS;
        f3 dir = mk(0, 0, 0), p = mk(0, 0, 0), p_out = mk(0, 0, 0), p_in = mk(0, 0, 0), nrm = mk(0, 0, 0);
        f3 diffuse = mk(0, 0, 0), beta = mk(1, 1, 1);
        if (active) {
            slot = q_slot_in[i];
            f3 pos = mk(r[i], r[i + cap], r[i + 2 * cap]);
            dir = mk(r[i + 3 * cap], r[i + 4 * cap], r[i + 5 * cap]);
            if (next.enabled) {   // this entry's primary ray of the next pass (same origin: only the direction changes)
                const uint2 px = W.p_pix[i];
                const f3 nd = primary_dir(next.cam, px.x, px.y, next.si, next.sj, next.aa);
                W.p_ray[i + 3 * cap] = nd.x; W.p_ray[i + 4 * cap] = nd.y; W.p_ray[i + 5 * cap] = nd.z;
            }
            if (fold) {
                if (fold == 2u) {   // (the pending skybox term is the sample's last)
                    W.sum[slot] = W.sum[slot] + (W.L[slot] + W.Lp[slot]);
                    W.sum[slot + cap] = W.sum[slot + cap] + (W.L[slot + cap] + W.Lp[slot + cap]);
                    W.sum[slot + 2 * cap] = W.sum[slot + 2 * cap] + (W.L[slot + 2 * cap] + W.Lp[slot + 2 * cap]);
                }
                W.L[slot] = 0.0f; W.L[slot + cap] = 0.0f; W.L[slot + 2 * cap] = 0.0f;
                W.Lp[slot] = 0.0f; W.Lp[slot + cap] = 0.0f; W.Lp[slot + 2 * cap] = 0.0f;
            } else {
                beta = mk(W.beta[slot], W.beta[slot + cap], W.beta[slot + 2 * cap]);
            }
            Best b;
            const uint4 rec0 = rec_in[2 * (size_t)i];
            b.obj = rec0.x;
            if (b.obj == MISS) {  // integrator.rs:66
                f3 c = beta * sky_color(S, dir);
                W.Lp[slot] = c.x; W.Lp[slot + cap] = c.y; W.Lp[slot + 2 * cap] = c.z;   // the path ends: its last term
            } else {
                hit = true;
                b.prim = rec0.y;
                b.t_local = __uint_as_float(rec0.z); b.u = __uint_as_float(rec0.w); b.v = __uint_as_float(rec_in[2 * (size_t)i + 1].x);
                f3 hp, hn;
                float hu, hv;
                reconstruct_hit(S, b, pos, dir, hp, hn, hu, hv);
                mat = S.objects[b.obj].material;
                nrm = normalize(hn);                                  // integrator.rs:72-76
                bool outside = dot(nrm, -dir) > 0.0f;
                p_out = hp + nrm * 0.001f;
                p_in = hp - nrm * 0.001f;
                p = outside ? p_out : p_in;
                diffuse = material_diffuse(S, S.materials[mat], hu, hv);
                lit = !is_zero(diffuse);                               // integrator.rs:80
            }
        }
        // Every lit hit emits the same number of shadow rays (`n_shoot`, the samples of the lights that can need one), so
        // the shadow-ray queue needs no compaction of its own: the block's rays go to [lbase * n_shoot, (lbase + lcnt) *
        // n_shoot), grouped by sample.
        SHADE_PROBE(0 + (__float_as_uint(diffuse.x) == 0x7fc12345u))   // the classification is done: waits for phase A's loads
        uint32_t lbase, lcnt;
        uint32_t lq = block_append_range<STAGE_BLOCK / 64>(lit, &W.counts[lit_idx], &W.counts[any_idx], n_shoot, sh_app, lbase, lcnt);
        SHADE_PROBE(2)
        // ---- phase B: draw light samples, then the scatter direction (same stream order) ----
        f3 next_pos = mk(0, 0, 0), next_dir = mk(0, 0, 0);
        Rng rng; rng.lo = 0; rng.hi = 0;
        if (hit) { rng.lo = W.rng_lo[slot]; rng.hi = W.rng_hi[slot]; }
        if (lit) {                                                     // integrator.rs:80-88
            T.lit_slot[lq] = slot;
            float* f = T.lit_f;
            f[lq] = beta.x; f[lq + cap] = beta.y; f[lq + 2 * cap] = beta.z;
            f[lq + 3 * cap] = diffuse.x; f[lq + 4 * cap] = diffuse.y; f[lq + 5 * cap] = diffuse.z;
        }
        {
            uint32_t s = 0, s_shoot = 0;
            for (uint32_t l = 0; l < S.n_lights; l++) {
                const DevLight lt = S.lights[l];
                const int ns = light_is_delta(lt) ? 1 : shadow_rays;
                const bool shoots = light_shoots(lt);
                for (int k = 0; k < ns; k++, s++) {
                    if (lit) {
                        f3 sd, ill;
                        float tm;
                        light_sample(lt, p, nrm, rng, sd, tm, ill);
                        // What the resolve stage cannot rebuild from the light alone: nothing for an ambient or an occlusion
                        // light (their t_max and illumination are constants of the light), the cosine for a direction light.
                        float* q = T.smp + (size_t)s * 4 * cap + lq;
                        if (lt.kind == LIGHT_POINT) { q[0] = tm; q[cap] = ill.x; q[2 * (size_t)cap] = ill.y; q[3 * (size_t)cap] = ill.z; }
                        else if (lt.kind == LIGHT_DIRECTION) q[0] = rust_max(dot(sd, nrm), 0.0f);   // the factor c of light.rs:120-122
                        // A sample with t_max = 0 of a light that can shoot (a point light at distance exactly 0) is queued
                        // as a zero-length ray, which nothing occludes; the resolve stage takes it as visible either way.
                        if (shoots) {
                            const size_t aq = (size_t)lbase * n_shoot + (size_t)s_shoot * lcnt + (lq - lbase);
                            float* a = T.a_ray;
                            a[aq] = p.x; a[aq + ac] = p.y; a[aq + 2 * ac] = p.z;
                            a[aq + 3 * ac] = sd.x; a[aq + 4 * ac] = sd.y; a[aq + 5 * ac] = sd.z;
                            a[aq + 6 * ac] = tm;
                            T.a_vidx[aq] = s * cap + lq;
                        }
                    }
                    if (shoots) s_shoot++;
                }
            }
        }
        SHADE_PROBE(3 + (rng.lo == 0x123456789abcull && lit))   // light samples drawn
        if (hit) {
            const DevMaterial m = S.materials[mat];
            f3 out, w;
            if (material_scatter(m, nrm, dir, rng, out, w)) {          // integrator.rs:90-101
                f3 p2 = dot(out, nrm) > 0.0f ? p_out : p_in;
                f3 nb = beta * w;
                if (depth + 1 >= max_depth) {                          // integrator.rs:59-61 at depth+1
                    f3 c = nb * sky_color(S, out);
                    // (it comes before this hit's own light in the sum: the resolve of this bounce adds it first)
                    W.Lp[slot] = c.x; W.Lp[slot + cap] = c.y; W.Lp[slot + 2 * cap] = c.z;
                } else {
                    cont = true;
                    next_pos = p2; next_dir = out;
                    W.beta[slot] = nb.x; W.beta[slot + cap] = nb.y; W.beta[slot + 2 * cap] = nb.z;
                }
            }
            W.rng_lo[slot] = rng.lo; W.rng_hi[slot] = rng.hi;
        }
        SHADE_PROBE(5 + (cont && rng.lo == 0x123456789abcull))      // scatter direction drawn
        // ---- compaction of the continuing paths into the next ray queue ----
        uint32_t nq = block_append<STAGE_BLOCK / 64>(cont, &W.counts[q_out_idx], sh_app);
        SHADE_PROBE(7)
        if (cont) {
            q_slot_out[nq] = slot;
            float* o = q_ray_out;
            o[nq] = next_pos.x; o[nq + cap] = next_pos.y; o[nq + 2 * cap] = next_pos.z;
            o[nq + 3 * cap] = next_dir.x; o[nq + 4 * cap] = next_dir.y; o[nq + 5 * cap] = next_dir.z;
        }
    }
}

// Marks every queued path as a miss (max_depth = 0: integrate_recur returns the skybox at once).
__global__ void k_mark_miss(WaveBufs W, uint32_t q_idx) {
    const uint32_t n = W.counts[q_idx];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) W.p_rec[2 * (size_t)i].x = MISS;
}

// ------------------------------------------------------------------------------------------------
// Stage 4: resolve the lit queue once its shadow rays are traced (illumination(),
// src/integrator.rs:107-132) and add `diffuse * illumination` (integrator.rs:87), weighted by the
// path throughput, to the sample's radiance.  Sums run in the reference's order.
// ------------------------------------------------------------------------------------------------
// (block `bid` of the `n_blocks` blocks that work on this lit queue; last: see ResolveJob)
__device__ __forceinline__ void resolve_part(const DevScene& S, const uint32_t cap, float* L, float* Lp, const ShadeSet& T, int32_t shadow_rays, const uint32_t n,
                                             const uint32_t bid, const uint32_t n_blocks, const uint32_t last) {
    for (uint32_t j = bid * blockDim.x + threadIdx.x; j < n; j += n_blocks * blockDim.x) {
        const float* f = T.lit_f;
        f3 beta = mk(f[j], f[j + cap], f[j + 2 * cap]);
        f3 diffuse = mk(f[j + 3 * cap], f[j + 4 * cap], f[j + 5 * cap]);
        f3 illum = fill(0.0f);
        uint32_t s = 0;
        for (uint32_t l = 0; l < S.n_lights; l++) {
            const DevLight lt = S.lights[l];
            int ns = light_is_delta(lt) ? 1 : shadow_rays;
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
                bool visible = (tm == 0.0f) || (T.vis[(size_t)s * cap + j] == 0);   // integrator.rs:124
                if (visible) total = total + ill;
            }
            illum = illum + total / (float)ns;             // integrator.rs:131, :84
        }
        f3 c = beta * (diffuse * illum);
        uint32_t slot = T.lit_slot[j];
        if (last) {   // the scatter ray of the last bounce left the skybox term of the sample's end: it precedes this hit's light
            const f3 l0 = mk(L[slot] + Lp[slot], L[slot + cap] + Lp[slot + cap], L[slot + 2 * cap] + Lp[slot + 2 * cap]);
            Lp[slot] = 0.0f; Lp[slot + cap] = 0.0f; Lp[slot + 2 * cap] = 0.0f;
            L[slot] = l0.x + c.x; L[slot + cap] = l0.y + c.y; L[slot + 2 * cap] = l0.z + c.z;
        } else {
            L[slot] += c.x; L[slot + cap] += c.y; L[slot + 2 * cap] += c.z;
        }
    }
}
__global__ __launch_bounds__(256) void k_wave_resolve(DevScene S, WaveBufs W, ShadeSet T, float* L, float* Lp, int32_t shadow_rays, uint32_t lit_idx, uint32_t last) {
    resolve_part(S, W.cap, L, Lp, T, shadow_rays, W.counts[lit_idx], blockIdx.x, gridDim.x, last);
}

__global__ __launch_bounds__(STAGE_BLOCK, SHADE_WAVES_PER_SIMD) void k_wave_shade(DevScene S, WaveBufs W, ShadeSet T, int32_t depth, int32_t max_depth, int32_t shadow_rays,
                                                    uint32_t fold, uint32_t q_in_idx, uint32_t q_out_idx, uint32_t lit_idx, uint32_t any_idx, uint32_t n_shoot,
                                                    NextPass next) {
    shade_part(S, W, T, depth, max_depth, shadow_rays, fold, q_in_idx, q_out_idx, lit_idx, any_idx, n_shoot, next, blockIdx.x, gridDim.x);
}
// The shade launch of a bounce and, on blocks of their own behind the shading ones, up to two resolve jobs (the lit hits of the
// bounce before it; with bounce 1 also those of the previous pass's last bounce).  The jobs write radiance buffers and read
// shade sets that the shading blocks do not touch (WaveBufs::Lp, WaveBufs::set), so the parts need no order among them.
struct ResolveJobDev {
    float* L;
    float* Lp;
    ShadeSet T;
    uint32_t lit_idx, last;
};
__global__ __launch_bounds__(STAGE_BLOCK, SHADE_WAVES_PER_SIMD) void k_wave_shade_resolve(DevScene S, WaveBufs W, ShadeSet T, int32_t depth, int32_t max_depth, int32_t shadow_rays,
                                                            uint32_t q_in_idx, uint32_t q_out_idx, uint32_t lit_idx, uint32_t any_idx, uint32_t n_shoot,
                                                            uint32_t shade_blocks, uint32_t job_blocks, ResolveJobDev j0, ResolveJobDev j1) {
    if (blockIdx.x < shade_blocks) {
        NextPass none;
        none.enabled = 0u; none.si = 0u; none.sj = 0u; none.aa = 0u; none.zero_base = MISS;
        shade_part(S, W, T, depth, max_depth, shadow_rays, 0u, q_in_idx, q_out_idx, lit_idx, any_idx, n_shoot, none, blockIdx.x, shade_blocks);
        return;
    }
    const uint32_t b = blockIdx.x - shade_blocks;
    if (b < job_blocks) resolve_part(S, W.cap, j0.L, j0.Lp, j0.T, shadow_rays, W.counts[j0.lit_idx], b, job_blocks, j0.last);
    else if (j1.lit_idx != MISS) resolve_part(S, W.cap, j1.L, j1.Lp, j1.T, shadow_rays, W.counts[j1.lit_idx], b - job_blocks, job_blocks, j1.last);
}

// ------------------------------------------------------------------------------------------------
// Stage 5: pixel = mean of the samples (integrator.rs:50-53), written to the tile-major output.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(WaveBufs W, RegionDev R, uint32_t chunk_begin, uint32_t chunk_n, float n_samples, const float* L1,
                                                  const float* Lp1, const float* L2, const float* Lp2, float* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chunk_n) return;
    uint32_t cx, cy;
    bool valid = slot_pixel(R, chunk_begin + i, cx, cy);
    size_t o = 3 * slot_out(R, chunk_begin + i);
    if (valid) {
        // the samples are added in generation order (integrator.rs:50-53): the sum, then the last two passes
#pragma unroll
        for (uint32_t c = 0; c < 3; c++) {
            float v = W.sum[i + c * W.cap];
            if (L1 != nullptr) v = v + (L1[i + c * W.cap] + Lp1[i + c * W.cap]);
            v = v + (L2[i + c * W.cap] + Lp2[i + c * W.cap]);
            out[o + c] = v / n_samples;
        }
    } else {
        out[o] = 0.0f; out[o + 1] = 0.0f; out[o + 2] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// Depth-first form of the integrator for scatter_rays != 1.  The reference's recursion draws the random numbers of a
// whole scatter subtree before the next scatter ray of the same hit (integrator.rs:90-102), an order that a
// breadth-first wavefront cannot reproduce; here one lane owns one pixel and walks its path tree with an explicit
// stack, so the stream order and the nested summation `color += scatter * L / scatter_rays` are the reference's.
// The stack lives in HBM, component-major: word f of the frame of depth d of pixel slot i at [(d * FRAME + f) * cap + i].
// ------------------------------------------------------------------------------------------------
struct PathFrame {
    static constexpr uint32_t WORDS = 20;
    f3 color, w, n, dir, p_out, p_in;
    uint32_t mat;
    int32_t k;   // scatter rays drawn so far
};
__device__ __forceinline__ void frame_store(float* stack, size_t cap, uint32_t i, int d, const PathFrame& f) {
    float* q = stack + (size_t)d * PathFrame::WORDS * cap + i;
    const float v[PathFrame::WORDS] = {f.color.x, f.color.y, f.color.z, f.w.x, f.w.y, f.w.z, f.n.x, f.n.y, f.n.z, f.dir.x, f.dir.y, f.dir.z,
                                       f.p_out.x, f.p_out.y, f.p_out.z, f.p_in.x, f.p_in.y, f.p_in.z, __uint_as_float(f.mat), __int_as_float(f.k)};
#pragma unroll
    for (uint32_t c = 0; c < PathFrame::WORDS; c++) q[c * cap] = v[c];
}
__device__ __forceinline__ void frame_load(const float* stack, size_t cap, uint32_t i, int d, PathFrame& f) {
    const float* q = stack + (size_t)d * PathFrame::WORDS * cap + i;
    float v[PathFrame::WORDS];
#pragma unroll
    for (uint32_t c = 0; c < PathFrame::WORDS; c++) v[c] = q[c * cap];
    f.color = mk(v[0], v[1], v[2]); f.w = mk(v[3], v[4], v[5]); f.n = mk(v[6], v[7], v[8]); f.dir = mk(v[9], v[10], v[11]);
    f.p_out = mk(v[12], v[13], v[14]); f.p_in = mk(v[15], v[16], v[17]);
    f.mat = __float_as_uint(v[18]); f.k = __float_as_int(v[19]);
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_pixel_depth_first(DevScene S, RegionDev R, CameraDev cam, int32_t max_depth, int32_t shadow_rays, int32_t scatter_rays,
                                                           uint32_t aa, uint32_t chunk_begin, uint32_t chunk_n, float* stack, uint32_t cap, float* out,
                                                           unsigned long long* stats) {
    LdsMem mem;
    const LdsMem* lm = stage_mesh<USE_LDS>(S, mem);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_closest = 0, n_any = 0;
    if (i < chunk_n) {
        uint32_t cx = 0, cy = 0;
        const bool valid = slot_pixel(R, chunk_begin + i, cx, cy);
        f3 total = fill(0.0f);
        if (valid) {
            Rng rng = rng_seed((uint64_t)cx ^ __builtin_bswap64((uint64_t)cy));  // integrator.rs:36
            for (uint32_t pass = 0; pass < aa * aa; pass++) {
                const uint32_t si = pass / aa, sj = pass % aa;
                const float n = (float)aa;
                const float x = (float)cx + ((float)si + 0.5f) / n - 0.5f;  // integrator.rs:41-42
                const float y = (float)cy + ((float)sj + 0.5f) / n - 0.5f;
                const float u = 2.0f * (x / (float)cam.width) - 1.0f;       // scene.rs:84-90
                const float v = 2.0f * (y / (float)cam.height) - 1.0f;
                f3 rpos = cam.pos;
                f3 rdir = normalize((cam.dir + u * cam.horizontal) + v * cam.vertical);
                enum { ENTER, SCATTER, RETURN };
                int state = ENTER, d = 0;
                f3 ret = fill(0.0f);
                PathFrame cur;
                cur.color = cur.w = cur.n = cur.dir = cur.p_out = cur.p_in = fill(0.0f); cur.mat = 0; cur.k = 0;
                for (;;) {
                    if (state == ENTER) {               // integrate_recur(ray, d), integrator.rs:58-88
                        bool ends = true;
                        if (d < max_depth) {
                            Best b;
                            b.obj = MISS; b.prim = 0; b.t_local = 0.f; b.u = 0.f; b.v = 0.f;
                            float t_hit;
                            n_closest++;
                            if (scene_closest(S, lm, rpos, rdir, 1e12f, b, t_hit)) {
                                ends = false;
                                f3 hp, hn;
                                float hu, hv;
                                reconstruct_hit(S, b, rpos, rdir, hp, hn, hu, hv);
                                cur.mat = S.objects[b.obj].material;
                                cur.n = normalize(hn);
                                const bool outside = dot(cur.n, -rdir) > 0.0f;
                                cur.p_out = hp + cur.n * 0.001f;
                                cur.p_in = hp - cur.n * 0.001f;
                                cur.dir = rdir;
                                cur.color = fill(0.0f);
                                cur.k = 0;
                                const f3 p = outside ? cur.p_out : cur.p_in;
                                const f3 diffuse = material_diffuse(S, S.materials[cur.mat], hu, hv);
                                if (!is_zero(diffuse)) {
                                    f3 illum = fill(0.0f);
                                    for (uint32_t l = 0; l < S.n_lights; l++) {   // illumination(), integrator.rs:107-132
                                        const DevLight lt = S.lights[l];
                                        const int ns = light_is_delta(lt) ? 1 : shadow_rays;
                                        f3 tot = fill(0.0f);
                                        for (int k = 0; k < ns; k++) {
                                            f3 sd, ill;
                                            float tm;
                                            light_sample(lt, p, cur.n, rng, sd, tm, ill);
                                            bool visible = tm == 0.0f;
                                            if (!visible) {
                                                n_any++;
                                                visible = !scene_any(S, lm, p, sd, tm);
                                            }
                                            if (visible) tot = tot + ill;
                                        }
                                        illum = illum + tot / (float)ns;
                                    }
                                    cur.color = cur.color + diffuse * illum;
                                }
                                state = SCATTER;
                            }
                        }
                        if (ends) {                     // integrator.rs:59-61, :66
                            ret = sky_color(S, rdir);
                            state = RETURN;
                        }
                    } else if (state == SCATTER) {      // integrator.rs:90-102
                        bool descend = false;
                        while (cur.k < scatter_rays) {
                            cur.k++;
                            f3 o, w;
                            if (material_scatter(S.materials[cur.mat], cur.n, cur.dir, rng, o, w)) {
                                cur.w = w;
                                rpos = dot(o, cur.n) > 0.0f ? cur.p_out : cur.p_in;
                                rdir = o;
                                frame_store(stack, cap, i, d, cur);
                                d++;
                                state = ENTER;
                                descend = true;
                                break;
                            }
                        }
                        if (!descend) {
                            ret = cur.color;
                            state = RETURN;
                        }
                    } else {                            // the call at depth d returned `ret`
                        if (d == 0) break;
                        d--;
                        frame_load(stack, cap, i, d, cur);
                        cur.color = cur.color + (cur.w * ret) / (float)scatter_rays;
                        state = SCATTER;
                    }
                }
                total = total + ret;                    // integrator.rs:50-53
            }
            total = total / (float)(aa * aa);
        }
        const size_t o = 3 * slot_out(R, chunk_begin + i);
        out[o] = total.x; out[o + 1] = total.y; out[o + 2] = total.z;
    }
    // ray counts: one pair of atomics per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_closest += __shfl_down(n_closest, off);
        n_any += __shfl_down(n_any, off);
    }
    if ((threadIdx.x & 63u) == 0 && (n_closest | n_any)) {
        atomicAdd(&stats[0], n_closest);
        atomicAdd(&stats[1], n_any);
    }
}

// ------------------------------------------------------------------------------------------------
// The same depth-first integrator as a pipeline of rounds, for chunks large enough to fill the GPU.  A pixel's rays are
// inherently sequential (one random stream per pixel, and the number of draws a subtree takes depends on what its rays
// hit), but the pixels are independent: every pixel is a coroutine that runs until it needs the result of a closest-hit
// ray, suspends with that ray in a queue, and all suspended pixels' rays are traced by ONE launch of the scheduler
// kernel (k_wave_trace), which is an order of magnitude faster per ray than the nested loops of k_pixel_depth_first.
// The shadow rays of a hit do not suspend the pixel: illumination() (integrator.rs:107-132) draws all its light samples
// before the scatter direction, visibility only enters `color = diffuse * illum`, so the samples are kept, their rays
// ride with the pixel's next closest-hit ray, and the frame's colour is completed when the pixel resumes.  Only where
// that colour is needed at once (no scatter ray left, or a child that ends at the depth limit without a ray) does the
// pixel wait one round on its shadow rays alone.  Operation order per pixel is that of k_pixel_depth_first, so both
// produce the same bits.
//
// Per pixel slot (arrays of the wavefront workspace, reused): random stream, running pixel sum (W.sum), a suspended
// return value (W.Lb[0]), the pending hit's diffuse colour (W.beta) and light samples (set 0: smp = t_max, ill.xyz per
// sample; vis), meta word (W.p_pix: x = sample pass, y = depth | state << 8 | (pending depth + 1) << 16), index of its
// ray in the round's queue (W.p_slot).  Queues, double buffered by round parity: active pixels (W.q_slot), closest-hit
// rays (W.q_ray -> W.h_rec); shadow rays in set 0.  Counters: three sets of CounterSlots::TOTAL in rotation, round r
// reads set r % 3 (Q + 1 = active pixels, Q = closest-hit rays, ANY = shadow rays, HEAD = the trace launch's queue
// head), appends to the next and zeroes the one after.
// ------------------------------------------------------------------------------------------------
struct DfParams {
    RegionDev R;
    CameraDev cam;
    uint32_t chunk_begin, aa;
    int32_t max_depth, shadow_rays, scatter_rays;
    uint32_t n_shoot;
    float* stack;   // max_depth * PathFrame::WORDS * stack_cap
    float* out;
    uint32_t stack_cap;   // slots the frame stack is laid out for: the chunk size of this render, which can be smaller than the
                          // capacity of a workspace that an earlier, larger render left behind (W.cap)
};
enum : uint32_t { DF_ENTER = 0, DF_SCATTER = 1, DF_FOLD = 2 };

__global__ __launch_bounds__(DF_BLOCK) void k_df_begin(WaveBufs W, DfParams X, uint32_t chunk_n, uint32_t out_base) {
    __shared__ uint32_t sh_app[DF_BLOCK / 64 + 2];
    typedef CounterSlots CS;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cx = 0, cy = 0;
    const bool valid = i < chunk_n && slot_pixel(X.R, X.chunk_begin + i, cx, cy);
    uint32_t base, cnt;
    const uint32_t q = block_append_range<DF_BLOCK / 64>(valid, &W.counts[out_base + CS::Q + 1], &W.counts[out_base + CS::Q], 1u, sh_app, base, cnt);
    if (valid) {
        const Rng r = rng_seed((uint64_t)cx ^ __builtin_bswap64((uint64_t)cy));  // integrator.rs:36
        W.rng_lo[i] = r.lo; W.rng_hi[i] = r.hi;
        W.sum[i] = 0.0f; W.sum[i + W.cap] = 0.0f; W.sum[i + 2 * W.cap] = 0.0f;
        W.p_pix[i] = make_uint2(0u, 0u | (DF_ENTER << 8));
        W.p_slot[i] = q;
        W.q_slot[0][q] = i;
        const f3 dir = primary_dir(X.cam, cx, cy, 0u, 0u, X.aa);
        float* o = W.q_ray[0];
        const uint32_t cap = W.cap;
        o[q] = X.cam.pos.x; o[q + cap] = X.cam.pos.y; o[q + 2 * cap] = X.cam.pos.z;
        o[q + 3 * cap] = dir.x; o[q + 4 * cap] = dir.y; o[q + 5 * cap] = dir.z;
    } else if (i < chunk_n) {
        const size_t o = 3 * slot_out(X.R, X.chunk_begin + i);
        X.out[o] = 0.0f; X.out[o + 1] = 0.0f; X.out[o + 2] = 0.0f;
    }
}

__global__ __launch_bounds__(DF_BLOCK) void k_df_step(DevScene S, WaveBufs W, DfParams X, uint32_t cur, uint32_t in_base, uint32_t out_base, uint32_t zero_base) {
    __shared__ uint32_t sh_app[DF_BLOCK / 64 + 2];
    typedef CounterSlots CS;
    const uint32_t n = W.counts[in_base + CS::Q + 1];
    if (blockIdx.x == 0)   // nothing reads or appends to that set before this launch is over
        for (uint32_t k = threadIdx.x; k < CS::TOTAL; k += blockDim.x) W.counts[zero_base + k] = 0u;
    const uint32_t cap = W.cap;
    const size_t ac = W.any_cap;
    const uint32_t* act_in = W.q_slot[cur];
    uint32_t* act_out = W.q_slot[cur ^ 1u];
    const float* r_in = W.q_ray[cur];
    float* r_out = W.q_ray[cur ^ 1u];
    const ShadeSet T = W.set[0];
    float* ret_buf = W.Lb[0];
    const float n_scatter = (float)X.scatter_rays;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_iter = (n + stride - 1) / stride;
    for (uint32_t it = 0; it < n_iter; it++) {
        const uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool active = i < n;
        uint32_t slot = 0, pass = 0, pend = 0;
        int d = 0;
        Rng rng; rng.lo = 0; rng.hi = 0;
        PathFrame f;
        f.color = f.w = f.n = f.dir = f.p_out = f.p_in = fill(0.0f); f.mat = 0; f.k = 0;
        f3 ret = fill(0.0f), rpos = fill(0.0f), rdir = mk(0, 0, 1), diffuse = fill(0.0f), p_lit = fill(0.0f);
        bool lit = false, in_scatter = false, in_return = false;
        if (active) {
            slot = act_in[i];
            const uint2 meta = W.p_pix[slot];
            pass = meta.x;
            d = (int)(meta.y & 0xFFu);
            const uint32_t state = (meta.y >> 8) & 0xFFu;
            const uint32_t pend_in = meta.y >> 16;
            rng.lo = W.rng_lo[slot]; rng.hi = W.rng_hi[slot];
            if (pend_in) {   // the shadow rays of the pixel's last hit are back: illumination(), integrator.rs:107-132, then :87
                const f3 df = mk(W.beta[slot], W.beta[slot + cap], W.beta[slot + 2 * cap]);
                f3 illum = fill(0.0f);
                uint32_t s = 0;
                for (uint32_t l = 0; l < S.n_lights; l++) {
                    const DevLight lt = S.lights[l];
                    const int ns = light_is_delta(lt) ? 1 : X.shadow_rays;
                    f3 tot = fill(0.0f);
                    for (int k = 0; k < ns; k++, s++) {
                        const float* q = T.smp + (size_t)s * 4 * cap + slot;
                        const bool visible = (q[0] == 0.0f) || (T.vis[(size_t)s * cap + slot] == 0);   // integrator.rs:124
                        if (visible) tot = tot + mk(q[cap], q[2 * (size_t)cap], q[3 * (size_t)cap]);
                    }
                    illum = illum + tot / (float)ns;
                }
                const f3 c = fill(0.0f) + df * illum;
                float* q = X.stack + (size_t)(pend_in - 1u) * PathFrame::WORDS * X.stack_cap + slot;
                q[0] = c.x; q[X.stack_cap] = c.y; q[2 * (size_t)X.stack_cap] = c.z;   // (the frame's colour: its first three words)
            }
            if (state == DF_ENTER) {               // the closest-hit ray of integrate_recur(ray, d) is back, integrator.rs:64
                const uint32_t qi = W.p_slot[slot];
                rpos = mk(r_in[qi], r_in[qi + cap], r_in[qi + 2 * cap]);
                rdir = mk(r_in[qi + 3 * cap], r_in[qi + 4 * cap], r_in[qi + 5 * cap]);
                const uint4 rec0 = W.h_rec[2 * (size_t)qi];
                if (rec0.x == MISS) {              // integrator.rs:66
                    ret = sky_color(S, rdir);
                    in_return = true;
                } else {
                    Best b;
                    b.obj = rec0.x; b.prim = rec0.y;
                    b.t_local = __uint_as_float(rec0.z); b.u = __uint_as_float(rec0.w); b.v = __uint_as_float(W.h_rec[2 * (size_t)qi + 1].x);
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
                frame_load(X.stack, X.stack_cap, slot, d, f);
                if (state == DF_FOLD) {            // a scatter ray's radiance waited for this frame's colour, integrator.rs:98
                    ret = mk(ret_buf[slot], ret_buf[slot + cap], ret_buf[slot + 2 * cap]);
                    f.color = f.color + (f.w * ret) / n_scatter;
                }
                in_scatter = true;
            }
        }
        // every lit hit emits the same number of shadow rays, so their queue positions follow from the block's range
        uint32_t lbase, lcnt;
        const uint32_t lq = block_append_range<DF_BLOCK / 64>(lit, &W.counts[out_base + CS::LIT], &W.counts[out_base + CS::ANY], X.n_shoot, sh_app, lbase, lcnt);
        if (lit) {
            pend = (uint32_t)d + 1u;
            W.beta[slot] = diffuse.x; W.beta[slot + cap] = diffuse.y; W.beta[slot + 2 * cap] = diffuse.z;
            uint32_t s = 0, s_shoot = 0;
            for (uint32_t l = 0; l < S.n_lights; l++) {
                const DevLight lt = S.lights[l];
                const int ns = light_is_delta(lt) ? 1 : X.shadow_rays;
                const bool shoots = light_shoots(lt);
                for (int k = 0; k < ns; k++, s++) {
                    f3 sd, ill;
                    float tm;
                    light_sample(lt, p_lit, f.n, rng, sd, tm, ill);
                    float* q = T.smp + (size_t)s * 4 * cap + slot;
                    q[0] = tm; q[cap] = ill.x; q[2 * (size_t)cap] = ill.y; q[3 * (size_t)cap] = ill.z;
                    if (shoots) {   // (a sample with t_max = 0 is queued as a zero-length ray; the resolve takes it as visible either way)
                        const size_t aq = (size_t)lbase * X.n_shoot + (size_t)s_shoot * lcnt + (lq - lbase);
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
        bool want_ray = false, done = false;
        uint32_t new_state = DF_ENTER;
        if (active) {
            for (;;) {
                if (in_scatter) {                                   // integrator.rs:90-102
                    in_scatter = false;
                    bool descend = false;
                    while (f.k < X.scatter_rays) {
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
                        frame_store(X.stack, X.stack_cap, slot, d, f);
                        d++;
                        if (d < X.max_depth) { want_ray = true; new_state = DF_ENTER; break; }
                        ret = sky_color(S, rdir);                   // integrator.rs:59-61: the depth limit ends the path without a ray
                        in_return = true;
                    } else {
                        if (pend == (uint32_t)d + 1u) { frame_store(X.stack, X.stack_cap, slot, d, f); new_state = DF_SCATTER; break; }
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
                        if (pass < X.aa * X.aa) {
                            W.sum[slot] = total.x; W.sum[slot + cap] = total.y; W.sum[slot + 2 * cap] = total.z;
                            uint32_t cx = 0, cy = 0;
                            (void)slot_pixel(X.R, X.chunk_begin + slot, cx, cy);
                            rpos = X.cam.pos;
                            rdir = primary_dir(X.cam, cx, cy, pass / X.aa, pass % X.aa, X.aa);
                            want_ray = true; new_state = DF_ENTER;
                        } else {
                            total = total / (float)(X.aa * X.aa);
                            const size_t o = 3 * slot_out(X.R, X.chunk_begin + slot);
                            X.out[o] = total.x; X.out[o + 1] = total.y; X.out[o + 2] = total.z;
                            done = true;
                        }
                        break;
                    }
                    d--;
                    if (pend == (uint32_t)d + 1u) {                 // that frame's colour is still waiting for its shadow rays
                        ret_buf[slot] = ret.x; ret_buf[slot + cap] = ret.y; ret_buf[slot + 2 * cap] = ret.z;
                        new_state = DF_FOLD;
                        break;
                    }
                    frame_load(X.stack, X.stack_cap, slot, d, f);
                    f.color = f.color + (f.w * ret) / n_scatter;
                    in_scatter = true;
                }
            }
        }
        const bool still = active && !done;
        const uint32_t ai = block_append<DF_BLOCK / 64>(still, &W.counts[out_base + CS::Q + 1], sh_app);
        const uint32_t qi = block_append<DF_BLOCK / 64>(want_ray, &W.counts[out_base + CS::Q], sh_app);
        if (still) {
            act_out[ai] = slot;
            W.p_pix[slot] = make_uint2(pass, (uint32_t)d | (new_state << 8) | (pend << 16));
            W.rng_lo[slot] = rng.lo; W.rng_hi[slot] = rng.hi;
        }
        if (want_ray) {
            W.p_slot[slot] = qi;
            r_out[qi] = rpos.x; r_out[qi + cap] = rpos.y; r_out[qi + 2 * cap] = rpos.z;
            r_out[qi + 3 * cap] = rdir.x; r_out[qi + 4 * cap] = rdir.y; r_out[qi + 5 * cap] = rdir.z;
        }
    }
}

__global__ void k_quantize(const float* rgb, size_t n, uint8_t* out) {  // src/render.rs:62-72
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float f = floorf(rgb[i] * 256.0f);
    f = rust_min(rust_max(f, 0.0f), 255.0f);
    out[i] = (uint8_t)f;
}

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t trace_grid(const LaunchCfg& cfg, size_t n) {
    size_t need = (n + TRACE_BLOCK - 1) / TRACE_BLOCK;
    size_t cap = (size_t)cfg.num_cus * cfg.trace_blocks_per_cu;
    size_t g = need < cap ? need : cap;
    return (uint32_t)(g ? g : 1);
}

static inline uint32_t sched_grid(const LaunchCfg& cfg, size_t upper, bool no_mesh = false) {
    // a queue too short to give every wave of the full grid 64 rays is spread thinner, down to min_batch rays per wave
    // (sched_batch_size), the first wave of every block before any block's second: as many blocks as such batches
    const size_t mb = (size_t)(cfg.min_batch < 1 ? 1 : (cfg.min_batch > 64 ? 64 : cfg.min_batch));
    size_t need = (upper + mb - 1) / mb;
    // scenes without meshes run the leaner kernel instance at a higher occupancy: half as many blocks again per CU
    size_t cap = (size_t)cfg.num_cus * (no_mesh ? (cfg.sched_blocks_per_cu * NOMESH_WAVES_PER_SIMD + SCHED_WAVES_PER_SIMD - 1) / SCHED_WAVES_PER_SIMD : cfg.sched_blocks_per_cu);
    size_t g = need < cap ? need : cap;
    return (uint32_t)(g ? g : 1);
}
static inline uint32_t sched_chunk(const LaunchCfg& cfg, size_t n, uint32_t grid) {
    uint32_t chunk = (uint32_t)cfg.chunk;
    const size_t waves = (size_t)grid * (TRACE_BLOCK / 64);
    while (chunk > 64 && (n / chunk) < waves * 4) chunk >>= 1;
    return chunk | ((uint32_t)cfg.static_share << 16) | ((uint32_t)cfg.min_batch << 20) | ((uint32_t)(cfg.thin_tail != 0) << 28);
}

hipError_t launch_trace_closest(const DevScene& S, const LaunchCfg& cfg, const RayBatch& rays, const HitBatch& out, size_t n, uint32_t* work_head, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const bool lds = cfg.use_lds && S.lds_mesh != MISS;
    if (cfg.scheduler) {
        hipError_t e = hipMemsetAsync(work_head, 0, 4, st);
        if (e != hipSuccess) return e;
        uint32_t g = sched_grid(cfg, n, S.n_meshes == 0);
        BatchIO io{S, rays, out, nullptr, false};
        uint32_t chunk = sched_chunk(cfg, n, g);
        RTAMD_DISPATCH_MESHLOC(S, cfg, k_trace_batch_sched, g, st, S, io, (uint32_t)n, work_head, chunk);
        return hipGetLastError();
    }
    uint32_t g = trace_grid(cfg, n);
    if (lds) hipLaunchKernelGGL(k_trace_closest<true>, dim3(g), dim3(TRACE_BLOCK), S.lds_bytes, st, S, rays, out, (uint32_t)n);
    else hipLaunchKernelGGL(k_trace_closest<false>, dim3(g), dim3(TRACE_BLOCK), 0, st, S, rays, out, (uint32_t)n);
    return hipGetLastError();
}
hipError_t launch_trace_any(const DevScene& S, const LaunchCfg& cfg, const RayBatch& rays, uint8_t* out, size_t n, uint32_t* work_head, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const bool lds = cfg.use_lds && S.lds_mesh != MISS;
    if (cfg.scheduler) {
        hipError_t e = hipMemsetAsync(work_head, 0, 4, st);
        if (e != hipSuccess) return e;
        uint32_t g = sched_grid(cfg, n, S.n_meshes == 0);
        BatchIO io{S, rays, HitBatch{nullptr, nullptr, nullptr, nullptr, nullptr}, out, true};
        uint32_t chunk = sched_chunk(cfg, n, g);
        RTAMD_DISPATCH_MESHLOC(S, cfg, k_trace_batch_sched, g, st, S, io, (uint32_t)n, work_head, chunk);
        return hipGetLastError();
    }
    uint32_t g = trace_grid(cfg, n);
    if (lds) hipLaunchKernelGGL(k_trace_any<true>, dim3(g), dim3(TRACE_BLOCK), S.lds_bytes, st, S, rays, out, (uint32_t)n);
    else hipLaunchKernelGGL(k_trace_any<false>, dim3(g), dim3(TRACE_BLOCK), 0, st, S, rays, out, (uint32_t)n);
    return hipGetLastError();
}
hipError_t launch_raygen(const WaveBufs& W, const RegionDev& R, const CameraDev& cam, uint32_t chunk_begin, uint32_t chunk_n,
                         uint32_t si, uint32_t sj, uint32_t aa, bool first, uint32_t p_idx, hipStream_t st) {
    uint32_t g = (chunk_n + STAGE_BLOCK - 1) / STAGE_BLOCK;
    hipLaunchKernelGGL(k_raygen, dim3(g), dim3(STAGE_BLOCK), 0, st, W, R, cam, chunk_begin, chunk_n, si, sj, aa, first ? 1u : 0u, p_idx);
    return hipGetLastError();
}
hipError_t launch_mark_miss(const WaveBufs& W, uint32_t q_idx, size_t upper, hipStream_t st) {
    uint32_t g = (uint32_t)((upper + 255) / 256);
    if (g > 2048) g = 2048;
    if (!g) g = 1;
    hipLaunchKernelGGL(k_mark_miss, dim3(g), dim3(256), 0, st, W, q_idx);
    return hipGetLastError();
}
hipError_t launch_wave_trace(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, uint32_t cur, uint32_t q_idx, uint32_t p_idx, uint32_t a0_idx,
                             uint32_t a1_idx, uint32_t a0_set, uint32_t a1_set, uint32_t head_idx, size_t upper, hipStream_t st) {
    uint32_t g = sched_grid(cfg, upper, S.n_meshes == 0);
    RTAMD_DISPATCH_MESHLOC(S, cfg, k_wave_trace, g, st, S, W, W.set[a0_set & 3u], W.set[a1_set & 3u], cur, q_idx, p_idx, a0_idx, a1_idx, head_idx,
                           (uint32_t)cfg.chunk | ((uint32_t)cfg.static_share << 16) | ((uint32_t)cfg.min_batch << 20) | ((uint32_t)(cfg.thin_tail != 0) << 28));
    return hipGetLastError();
}
hipError_t launch_wave_shade(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, uint32_t set, int depth, int max_depth, int shadow_rays, uint32_t fold,
                             uint32_t q_in, uint32_t q_out, uint32_t lit_idx, uint32_t any_idx, uint32_t n_shoot, const NextPass& next, size_t upper,
                             const ResolveJob* r0, const ResolveJob* r1, hipStream_t st) {
    size_t need = (upper + STAGE_BLOCK - 1) / STAGE_BLOCK;
    size_t capg = (size_t)cfg.num_cus * (2048 / STAGE_BLOCK);
    uint32_t g = (uint32_t)(need < capg ? need : capg);
    if (!g) g = 1;
    if (r0 == nullptr && r1 != nullptr) { r0 = r1; r1 = nullptr; }
    if (r0 == nullptr) {
        hipLaunchKernelGGL(k_wave_shade, dim3(g), dim3(STAGE_BLOCK), 0, st, S, W, W.set[set & 3u], depth, max_depth, shadow_rays, fold, q_in, q_out, lit_idx, any_idx,
                           n_shoot, next);
        return hipGetLastError();
    }
    // (the shade part of such a launch is never bounce 0: nothing to fold, no next pass to prepare)
    size_t capj = (size_t)cfg.num_cus * 2;
    const uint32_t jb = (uint32_t)(need < capj ? need : capj);
    const ResolveJobDev j0{r0->L, r0->Lp, W.set[r0->set & 3u], r0->lit_idx, r0->last};
    const ResolveJobDev j1 = r1 ? ResolveJobDev{r1->L, r1->Lp, W.set[r1->set & 3u], r1->lit_idx, r1->last} : ResolveJobDev{nullptr, nullptr, W.set[0], MISS, 0u};
    hipLaunchKernelGGL(k_wave_shade_resolve, dim3(g + jb * (r1 ? 2u : 1u)), dim3(STAGE_BLOCK), 0, st, S, W, W.set[set & 3u], depth, max_depth, shadow_rays, q_in, q_out,
                       lit_idx, any_idx, n_shoot, g, jb, j0, j1);
    return hipGetLastError();
}
hipError_t launch_wave_resolve(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, int shadow_rays, const ResolveJob& job, size_t upper, hipStream_t st) {
    size_t need = (upper + 255) / 256;
    size_t capg = (size_t)cfg.num_cus * 8;
    uint32_t g = (uint32_t)(need < capg ? need : capg);
    if (!g) g = 1;
    hipLaunchKernelGGL(k_wave_resolve, dim3(g), dim3(256), 0, st, S, W, W.set[job.set & 3u], job.L, job.Lp, shadow_rays, job.lit_idx, job.last);
    return hipGetLastError();
}
hipError_t launch_finalize(const WaveBufs& W, const RegionDev& R, uint32_t chunk_begin, uint32_t chunk_n, float n_samples, const float* L_first,
                           const float* Lp_first, const float* L_second, const float* Lp_second, float* out, hipStream_t st) {
    uint32_t g = (chunk_n + 255) / 256;
    hipLaunchKernelGGL(k_finalize, dim3(g), dim3(256), 0, st, W, R, chunk_begin, chunk_n, n_samples, L_first, Lp_first, L_second, Lp_second, out);
    return hipGetLastError();
}
hipError_t launch_pixel_depth_first(const DevScene& S, const LaunchCfg& cfg, const RegionDev& R, const CameraDev& cam, int max_depth, int shadow_rays,
                                    int scatter_rays, uint32_t aa, uint32_t chunk_begin, uint32_t chunk_n, float* stack, uint32_t cap, float* out,
                                    unsigned long long* stats, hipStream_t st) {
    if (chunk_n == 0) return hipSuccess;
    const bool lds = cfg.use_lds && S.lds_mesh != MISS;
    const uint32_t g = (chunk_n + 255) / 256;
    if (lds) hipLaunchKernelGGL(k_pixel_depth_first<true>, dim3(g), dim3(256), S.lds_bytes, st, S, R, cam, max_depth, shadow_rays, scatter_rays, aa, chunk_begin,
                                chunk_n, stack, cap, out, stats);
    else hipLaunchKernelGGL(k_pixel_depth_first<false>, dim3(g), dim3(256), 0, st, S, R, cam, max_depth, shadow_rays, scatter_rays, aa, chunk_begin,
                            chunk_n, stack, cap, out, stats);
    return hipGetLastError();
}
hipError_t launch_df_begin(const WaveBufs& W, const DfFrame& F, uint32_t chunk_begin, uint32_t chunk_n, uint32_t out_base, hipStream_t st) {
    if (chunk_n == 0) return hipSuccess;
    const DfParams X{F.R, F.cam, chunk_begin, F.aa, F.max_depth, F.shadow_rays, F.scatter_rays, F.n_shoot, F.stack, F.out, F.stack_cap};
    const uint32_t g = (chunk_n + DF_BLOCK - 1) / DF_BLOCK;
    hipLaunchKernelGGL(k_df_begin, dim3(g), dim3(DF_BLOCK), 0, st, W, X, chunk_n, out_base);
    return hipGetLastError();
}
hipError_t launch_df_step(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, const DfFrame& F, uint32_t chunk_begin, uint32_t cur, uint32_t in_base,
                          uint32_t out_base, uint32_t zero_base, size_t upper, hipStream_t st) {
    const DfParams X{F.R, F.cam, chunk_begin, F.aa, F.max_depth, F.shadow_rays, F.scatter_rays, F.n_shoot, F.stack, F.out, F.stack_cap};
    size_t need = (upper + DF_BLOCK - 1) / DF_BLOCK;
    size_t capg = (size_t)cfg.num_cus * (2048 / DF_BLOCK);
    uint32_t g = (uint32_t)(need < capg ? need : capg);
    if (!g) g = 1;
    hipLaunchKernelGGL(k_df_step, dim3(g), dim3(DF_BLOCK), 0, st, S, W, X, cur, in_base, out_base, zero_base);
    return hipGetLastError();
}
hipError_t launch_quantize(const float* rgb, size_t n, uint8_t* out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint32_t g = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(k_quantize, dim3(g), dim3(256), 0, st, rgb, n, out);
    return hipGetLastError();
}

#ifdef RTAMD_TIMELINE
// out: [0] earliest wave start, [1] latest start, [2] earliest moment a wave found the queue empty, [3] latest wave end (all of
// the last launch), [4] sum of wave lifetimes, [5] waves, [6] sum of lifetimes after the wave's queue ran dry, [7] loop trips
// (sums over the launches since the last reset).
hipError_t timeline_read(unsigned long long* out, int reset) {
    static std::vector<unsigned long long> tab(TL_MAX_WAVES * 8);
    hipError_t e = hipMemcpyFromSymbol(tab.data(), HIP_SYMBOL(g_timeline), tab.size() * 8);
    if (e != hipSuccess) return e;
    unsigned long long r[8] = {~0ull, 0, ~0ull, 0, 0, 0, 0, 0};
    for (uint32_t w = 0; w < TL_MAX_WAVES; w++) {
        const unsigned long long* t = tab.data() + 8 * (size_t)w;
        if (t[5] == 0) continue;
        if (t[0] < r[0]) r[0] = t[0];
        if (t[0] > r[1]) r[1] = t[0];
        if (t[1] && t[1] < r[2]) r[2] = t[1];
        if (t[2] > r[3]) r[3] = t[2];
        r[4] += t[3]; r[5] += t[5]; r[6] += t[4]; r[7] += t[6];
    }
    for (int k = 0; k < 8; k++) out[k] = r[k];
    if (reset) {
        std::fill(tab.begin(), tab.end(), 0ull);
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), tab.data(), tab.size() * 8);
    }
    return e;
}
#endif
#ifdef RTAMD_TIMELINE
// The per-launch stamp table (TL_LAUNCH_SLOTS x TL_MAX_WAVES x {start, dry, end, trips}); zeroed when reset.
hipError_t timeline_dump(unsigned long long* out, size_t words, int reset) {
    const size_t all = (size_t)TL_LAUNCH_SLOTS * TL_MAX_WAVES * 4;
    if (words < all) return hipErrorInvalidValue;
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline_launch), all * 8);
    if (e != hipSuccess || !reset) return e;
    std::vector<unsigned long long> z(all, 0ull);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_timeline_launch), z.data(), all * 8);
}
#endif
#ifdef RTAMD_DIAG
hipError_t diag_read_shade(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_shade_diag), 256);
    if (e != hipSuccess) return e;
    if (reset) {
        unsigned long long z[32] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_shade_diag), z, 256);
    }
    return e;
}
hipError_t diag_read(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag), 256);
    if (e != hipSuccess) return e;
    if (reset) {
        unsigned long long z[32] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, 256);
    }
    return e;
}
#endif

}  // namespace rtamd
