// Shared declarations between the kernel file and the C-ABI implementation.
#pragma once
#include "rt_device.h"

struct rtamd_bvh_node;

namespace rtamd {

static constexpr int TRACE_BLOCK = 512;  // threads per block of the traversal kernels (8 waves)
#ifndef RTAMD_STAGE_BLOCK
#define RTAMD_STAGE_BLOCK 512
#endif
// threads per block of the queue-building stages (one queue atomic per block).  The shade kernel is a long dependent chain with two
// block-wide appends: two 512-thread blocks per CU overlap one block's barrier and atomic with the other's arithmetic (shade +
// resolve 15.0 -> 14.5 ms per headline frame against one block of 1024; 256 costs more in queue atomics than it returns).
static constexpr int STAGE_BLOCK = RTAMD_STAGE_BLOCK;
// the round pipeline of scatter_rays != 1 (k_df_begin / k_df_step) appends three queues per step: one atomic each per 1024 threads
// (4K frame, scatter 2: 1 922 Mrays/s against 1 682 with blocks of 512)
static constexpr int DF_BLOCK = 1024;
#ifndef RTAMD_SHADE_WAVES
#define RTAMD_SHADE_WAVES 4
#endif
static constexpr int SHADE_WAVES_PER_SIMD = RTAMD_SHADE_WAVES;  // register budget of the shade kernel
#ifndef RTAMD_SCHED_WAVES
#define RTAMD_SCHED_WAVES 4
#endif
#ifndef RTAMD_NOMESH_WAVES
#define RTAMD_NOMESH_WAVES 6
#endif
// scheduler kernels of scenes without meshes (mesh phases compiled out): three 8-wave blocks per CU.  They need 74-82 VGPRs
// and fit the 80 of six waves per SIMD without a spill since the loop's phases are a sequence of ifs (before that: 95-101,
// 14 spilled at six, and two blocks per CU were faster).  10 000 spheres at 4K, 16 spp: 3 215 Mrays/s at 4, 3 560 at 5 and 6.
static constexpr int NOMESH_WAVES_PER_SIMD = RTAMD_NOMESH_WAVES;
static constexpr int SCHED_WAVES_PER_SIMD = RTAMD_SCHED_WAVES;  // register budget of the scheduler kernels: two 8-wave blocks per CU = 4 waves/SIMD, 128 VGPRs

struct RayBatch { const float *ox, *oy, *oz, *dx, *dy, *dz, *tmax; };
struct HitBatch { uint32_t *obj, *prim; float *t, *u, *v; };

struct CameraDev { f3 pos, dir, horizontal, vertical; uint32_t width, height; };

// Owned pixel set of one render call (see rtamd_region in rtamd.h).
struct RegionDev {
    uint32_t x0, y0, w, h;
    uint32_t tile_w, tile_h, tile_first, tile_stride;
    uint32_t tiles_x, n_tiles;
    uint32_t squares;   // order of the slots inside a tile: 1 = squares of 8 x 8 pixels (both sides of the tile multiples of 8), 0 = rows
};

struct ShadeSet {
    // lit queue: paths whose hit has a non-black diffuse colour
    uint32_t* lit_slot;   // cap
    float* lit_f;         // 6*cap: beta.xyz, diffuse.xyz
    float* smp;           // 4*S*cap: per light sample t_max, ill.xyz
    // shadow-ray queue
    float* a_ray;         // 7*any_cap: pos.xyz, dir.xyz, t_max
    uint32_t* a_vidx;     // any_cap: index into vis (sample * cap + lit index)
    uint8_t* vis;         // S*cap: 1 = occluded
};

// Wavefront state of one chunk of `cap` pixel slots, structure-of-arrays in HBM.
// Vectors are stored component-major: component c of element i lives at [c * cap + i], so a wave
// reads 64 consecutive floats per component.
struct WaveBufs {
    uint32_t cap;
    uint32_t any_cap;     // capacity of the shadow-ray queue = cap * light samples per lit hit
    // per pixel slot, persistent over the samples of the pixel
    uint64_t* rng_lo;     // SmallRng state, low / high half
    uint64_t* rng_hi;
    float* sum;           // 3*cap: sum of finished samples
    float* L;             // 3*cap: radiance of the sample the launch works on (one of Lb, set by the host per launch)
    float* Lb[2];         // the two radiance buffers: sample pass k accumulates into Lb[k & 1] while the last shadow rays
                          // of pass k-1 are still being resolved into the other one
    // 3*cap, one per radiance buffer: the skybox term that ends a sample's path (a ray that misses, or the scatter ray of the
    // last bounce).  The shade launch leaves it here instead of adding it to L, so that it writes nothing the resolve launch
    // of the bounce before it writes and the two can run side by side; it is added in the place the sum would have had it:
    // by the resolve of the last bounce's lit hit, else when the sample is folded into the pixel sum.
    float* Lp;
    float* Lpb[2];
    float* beta;          // 3*cap: path throughput
    // closest-hit ray queues of the bounces d >= 1, double buffered by bounce parity
    uint32_t* q_slot[2];  // cap: pixel slot of the path
    float* q_ray[2];      // 6*cap: pos.xyz, dir.xyz
    // primary rays (bounce 0) have a queue and hit records of their own: they depend on nothing but the pixel and
    // the sample index, so the primary rays of pass k+1 are traced together with the last bounce of pass k
    uint32_t* p_slot;     // cap
    float* p_ray;         // 6*cap
    uint4* p_rec;         // 2*cap
    uint2* p_pix;         // cap: pixel (cx, cy) of the queue entry; every valid pixel has one primary ray per pass, so the
                          // queue keeps its entries over the passes and only the ray directions are rewritten
    // One 32-byte record per path: {object index or MISS, primitive, t in the object's space, u} {v, -, -, -}.
    // Rays of a wave finish in no particular order, so their results are scattered writes; as one
    // record they touch one cache line per ray instead of five.
    uint4* h_rec;         // 2*cap
    // What a shade launch leaves for the trace and resolve launches after it.  Four sets: bounce d of sample pass k writes
    // set 2 * (k & 1) + (d & 1).  The shade launch of bounce d runs beside the resolve launch of bounce d - 1 (other bounce
    // parity), and the lit hits of a pass's last bounce are resolved during the next pass (other pass parity).  (The
    // depth-first rounds and the frame kernel use set 0 only.)
    ShadeSet set[4];
    // counters: three sets of CounterSlots::TOTAL (sample pass k uses set k % 3, zeroed before its primary rays are
    // generated, i.e. during pass k-1, while the last counters of pass k-2 may still be in use); kernels take absolute
    // indices into this array
    uint32_t* counts;
    uint64_t* stats;      // [0] closest-hit rays, [1] any-hit rays
};
// Layout of WaveBufs::counts.  Every bounce has its own slots, so nothing is reset inside a pass.
struct CounterSlots {
    static constexpr uint32_t MAX_DEPTH = 64;
    static constexpr uint32_t Q = 0;                    // Q + d   : paths entering bounce d
    static constexpr uint32_t LIT = Q + MAX_DEPTH + 1;  // LIT + d : lit hits of bounce d
    static constexpr uint32_t ANY = LIT + MAX_DEPTH;    // ANY + d : shadow rays of bounce d
    static constexpr uint32_t HEAD = ANY + MAX_DEPTH;   // HEAD + d: work-queue head of trace launch d
    static constexpr uint32_t PQ = HEAD + MAX_DEPTH + 1;  // primary rays of the pass
    static constexpr uint32_t TOTAL = PQ + 1;
    static constexpr uint32_t PQ_KEPT = 3 * TOTAL;        // behind the three sets: length of a primary queue that is kept over the passes
    static constexpr uint32_t WORDS = 3 * TOTAL + 1;
};
// What the shade launch of bounce 0 does for the pass after it (max_depth >= 2): rewrite the primary ray directions for
// that pass's sample and zero that pass's counter set.
struct NextPass {
    uint32_t enabled, si, sj, aa;
    uint32_t zero_base;   // first counter of the set to zero, or MISS
    CameraDev cam;
};

struct LaunchCfg {
    int num_cus = 256;
    int trace_blocks_per_cu = 4;   // grid-stride kernels
    int sched_blocks_per_cu = 2;   // persistent scheduler kernels (512 threads each): what fits a CU (LDS, registers)
    int chunk = 64;                // rays a wave claims per atomic from the shared part of a queue
    int static_share = 6;          // eighths of a queue dealt to the waves without atomics (0..8)
    int min_batch = 4;             // rays per wave below which a short queue is not spread any thinner (1..64)
    int use_lds = 1;
    int scheduler = 1;             // 1: persistent phase-voting scheduler, 0: plain nested-loop kernels (batch entry points only)
    int thin_tail = 1;             // 1: a queue of one to two batches per wave is dealt whole (a second, shorter batch per wave) instead of leaving its second part to the shared head
    int lean = 1;                  // 1: scenes of meshes + cuboids (or spheres + cuboids) under inline wrappers run the lean kernel form
};

hipError_t launch_trace_closest(const DevScene& S, const LaunchCfg& cfg, const RayBatch& rays, const HitBatch& out, size_t n, uint32_t* work_head, hipStream_t st);
hipError_t launch_trace_any(const DevScene& S, const LaunchCfg& cfg, const RayBatch& rays, uint8_t* out, size_t n, uint32_t* work_head, hipStream_t st);
// first: first sample of the pixel (seed its random stream, zero its sum)
hipError_t launch_raygen(const WaveBufs& W, const RegionDev& R, const CameraDev& cam, uint32_t chunk_begin, uint32_t chunk_n,
                         uint32_t si, uint32_t sj, uint32_t aa, bool first, uint32_t p_idx, hipStream_t st);
hipError_t launch_mark_miss(const WaveBufs& W, uint32_t q_idx, size_t upper, hipStream_t st);
// One launch over up to four queues (counter index MISS: absent): bounce rays q_ray[cur], primary rays, shadow rays of
// shade set a0_set (length counter a0_idx) and of shade set a1_set (a1_idx).
hipError_t launch_wave_trace(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, uint32_t cur, uint32_t q_idx, uint32_t p_idx,
                             uint32_t a0_idx, uint32_t a1_idx, uint32_t a0_set, uint32_t a1_set, uint32_t head_idx, size_t upper, hipStream_t st);
// A resolve job: the lit queue of shade set `set` (length counter lit_idx; MISS: no job) is resolved into the radiance
// buffer L; last != 0: the lit hits of a pass's last bounce, whose pixel may hold a pending skybox term in Lp (WaveBufs::Lp).
struct ResolveJob {
    float* L;
    float* Lp;
    uint32_t lit_idx, set, last;
};
// depth 0 shades the primary queue and starts the sample: fold = 2 adds the old content of W.L and W.Lp (the sample
// two passes back) to the pixel sum first, then W.L, W.Lp and the throughput are reset.  `set`: the shade set it fills.
// n_shoot: shadow rays per lit hit = light samples of the lights that can need one (not ambient, not reach 0).
// r0, r1: resolve jobs that run beside it in the same launch, on blocks of their own (null: none); they must not read the
// set the shade part fills.
hipError_t launch_wave_shade(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, uint32_t set, int depth, int max_depth, int shadow_rays, uint32_t fold,
                             uint32_t q_in, uint32_t q_out, uint32_t lit_idx, uint32_t any_idx, uint32_t n_shoot, const NextPass& next, size_t upper,
                             const ResolveJob* r0, const ResolveJob* r1, hipStream_t st);
hipError_t launch_wave_resolve(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, int shadow_rays, const ResolveJob& job, size_t upper, hipStream_t st);
// pixel = (sum + (L_first + Lp_first) + (L_second + Lp_second)) / n_samples; L_first may be null (a single sample pass)
hipError_t launch_finalize(const WaveBufs& W, const RegionDev& R, uint32_t chunk_begin, uint32_t chunk_n, float n_samples, const float* L_first,
                           const float* Lp_first, const float* L_second, const float* Lp_second, float* out, hipStream_t st);
// The whole integrator, depth first, one lane per pixel slot (any scatter_rays).  stack: max_depth * 20 * cap floats;
// out: the chunk's pixels (3 floats per slot, written for every slot); stats[0..1] += closest-hit / any-hit rays.
hipError_t launch_pixel_depth_first(const DevScene& S, const LaunchCfg& cfg, const RegionDev& R, const CameraDev& cam, int max_depth, int shadow_rays,
                                    int scatter_rays, uint32_t aa, uint32_t chunk_begin, uint32_t chunk_n, float* stack, uint32_t cap, float* out,
                                    unsigned long long* stats, hipStream_t st);
// The depth-first integrator as a pipeline of rounds (any scatter_rays, max_depth >= 1): every pixel of the chunk runs
// until it needs a closest-hit ray, one launch_wave_trace per round serves all of them.  Round r: launch_wave_trace over
// the queues of counter set r % 3 (q_idx = base + Q, a0_idx = base + ANY, head = base + HEAD, cur = r & 1), then
// launch_df_step(cur = r & 1, in = that set, out = the next, zero = the one after).  The frame is finished when a set's
// active count (base + Q + 1) is zero.  stack: max_depth * 20 * cap floats.
struct DfFrame {
    RegionDev R;
    CameraDev cam;
    uint32_t aa;
    int32_t max_depth, shadow_rays, scatter_rays;
    uint32_t n_shoot;   // shadow rays per lit hit
    float* stack;
    float* out;
    uint32_t stack_cap;   // slots per depth level of `stack`
};
hipError_t launch_df_begin(const WaveBufs& W, const DfFrame& F, uint32_t chunk_begin, uint32_t chunk_n, uint32_t out_base, hipStream_t st);
hipError_t launch_df_step(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, const DfFrame& F, uint32_t chunk_begin, uint32_t cur, uint32_t in_base,
                          uint32_t out_base, uint32_t zero_base, size_t upper, hipStream_t st);
// The integrator of scatter_rays = 1 with every wave owning a range of pixel slots from their first camera ray on (rt_mega.hip).
// A chunk holds at most 256 * mega_waves() slots (and W.cap); n_shoot (shadow rays per lit hit) at most 7, max_depth at least 1.
// launch_mega_init once per chunk, then launch_mega for consecutive windows [s_begin, s_end) of the pixels' samples up to
// antialiasing^2, each with the waves' units (mega_assign) in F.units / F.owner; launch_mega_group_cost leaves in F.group_cost
// what the next window's units are dealt by.
struct MegaFrame {
    RegionDev R;
    CameraDev cam;
    uint32_t aa;
    int32_t max_depth, shadow_rays;
    uint32_t n_shoot;
    uint32_t step_min;   // resumable pixels of a wave from which its integrator step runs without waiting for idle lanes (1..64)
    float* out;
    uint32_t* units;     // mega_banks() words per wave (device): mega_assign
    uint32_t* owner;     // one word per unit of 64 slots (device): mega_assign
    uint32_t* group_cost;   // one word (a float) per unit (device)
    int32_t scatter_rays;   // != 1: one launch_mega over all samples; needs `stack`
    float* stack;           // scatter_rays != 1: max_depth * 20 * W.cap floats (device)
    uint32_t* wave_scale;   // 2 * mega_waves() words (device)
};
uint32_t mega_waves(const DevScene& S, const LaunchCfg& cfg);
uint32_t mega_banks();   // units of 64 pixel slots a wave can hold
bool mega_supported(const DevScene& S, const LaunchCfg& cfg);
hipError_t launch_mega_init(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, const MegaFrame& F, uint32_t chunk_begin, uint32_t chunk_n, hipStream_t st);
hipError_t launch_mega(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, const MegaFrame& F, uint32_t chunk_begin, uint32_t chunk_n, uint32_t s_begin,
                       uint32_t s_end, hipStream_t st);
hipError_t launch_mega_group_cost(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, const MegaFrame& F, uint32_t chunk_n, hipStream_t st);
// cost: host copy of F.group_cost (null: the first deal); units: mega_banks() * mega_waves() words, owner: one per unit (host)
void mega_assign(const float* cost, uint32_t chunk_n, uint32_t n_waves, uint32_t* units, uint32_t* owner);
hipError_t launch_quantize(const float* rgb, size_t n, uint8_t* out, hipStream_t st);
// The reference's binned-SAH tree over n boxes (6 floats each, device memory), built on the GPU (rt_bvh.hip).
// nodes_out: room for 2 n records; *n_nodes_out includes the sentinel; order_out: n primitive indices in tree order.
hipError_t build_bvh_device(const float* boxes, uint32_t n, float hit_cost, ::rtamd_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* order_out, int num_cus,
                            hipStream_t st);
#ifdef RTAMD_TIMELINE
hipError_t timeline_read(unsigned long long* out, int reset);
hipError_t timeline_dump(unsigned long long* out, size_t words, int reset);
#endif
#ifdef RTAMD_DIAG
hipError_t diag_read(unsigned long long* out, int reset);
hipError_t diag_read_shade(unsigned long long* out, int reset);
hipError_t mega_diag_read(unsigned long long* out48, int reset);   // the frame kernel's counters (rt_sched.h: g_diag)
#endif

}  // namespace rtamd
