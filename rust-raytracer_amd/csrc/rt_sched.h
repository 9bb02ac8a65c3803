// Persistent wave-level ray scheduler: the traversal kernel body shared by the batched trace entry
// points and the wavefront integrator.
//
// Why: with one lane per ray and the reference's nested loops (objects -> nodes -> triangles) a
// wave64 spends most of its issue slots with a handful of live lanes (measured: 16 % lane
// utilisation on closest-hit rays, 9 % on shadow rays), because lanes miss an object's root box,
// sit in different loop levels, or finish early.  Here every lane runs a small state machine
// over the SAME per-ray operation sequence as the reference (so results stay bit-identical), and
// each trip of the wave's loop executes the one phase that the most lanes are waiting for:
//
//   FETCH  store the finished ray's result, start on a ray of the wave's staged batch
//   OBJ    fold a finished mesh walk, step the root list / object-tree leaf, enter the next object
//          (ray-space wrappers, 1/dir) or intersect a non-mesh primitive
//   TNODE  one node of an object tree (world space)
//   NODE   one node of a mesh tree (object space)
//   TRI    one triangle of the current leaf
//
// Lanes that run out of work are refilled from a batch of 64 consecutive rays that the wave keeps
// staged in its own 2 KB of LDS.  The batch is fetched with direct global->LDS loads
// (global_load_lds_dword, issued from inline asm so that the compiler's s_waitcnt bookkeeping does
// not see them) the moment the previous batch is used up, and is only waited for (one explicit
// s_waitcnt vmcnt(0)) in the later FETCH step that first reads it: the ray loads' latency overlaps
// the traversal phases, and no register holds a pending load.
#pragma once
#include <type_traits>
#include "rt_device.h"

namespace rtamd {

enum : uint32_t { ST_FETCH = 0, ST_OBJ = 1, ST_TNODE = 2, ST_NODE = 3, ST_TRI = 4, ST_DONE = 5, ST_WALK = 6 /* walks its mesh by itself */,
                  ST_IDLE = 7 /* per-frame kernel: wants a ray and found none ready; looks again in the next FETCH step */ };

// LDS copy of mesh S.lds_mesh, addressed through LDS-typed pointers so that the loads are
// ds_read_b128 (a generic pointer that may be LDS or global compiles to slow flat loads).
// Triangles: what the triangle test reads and nothing else, in three planes: {a.xyz, ab.x} and {ab.yz, ac.xy} as 16-byte
// records, ac.z as a float (the vertex ids of the 48-byte global record are the shade stage's): 36 bytes per triangle
// instead of 48, and the consecutive triangles of a leaf, read by consecutive lanes, lie in consecutive banks.
struct LdsMesh {
    lds_float4* nodes;
    lds_float4* tris;    // plane of the first quads
    lds_float4* tris1;   // plane of the second quads
    __attribute__((address_space(3))) const float* trisz;   // plane of ac.z
};

// Where the mesh trees of a launch live: decided on the host per scene.
// MESH_NONE: the scene has no mesh at all (spheres, cuboids, triangles only): the mesh phases are compiled out.
enum : int { MESH_GLOBAL = 0, MESH_LDS = 1, MESH_MIXED = 2, MESH_NONE = 3 };

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

static constexpr uint32_t STAY_NUM = 3, STAY_DEN = 4;
#ifndef RTAMD_NODE_STAY_NUM
#define RTAMD_NODE_STAY_NUM 1
#endif
#ifndef RTAMD_NODE_STAY_DEN
#define RTAMD_NODE_STAY_DEN 4
#endif
#ifndef RTAMD_OBJ_STAY_NUM
#define RTAMD_OBJ_STAY_NUM 2
#endif
static constexpr uint32_t OBJ_STAY_NUM = RTAMD_OBJ_STAY_NUM;
#ifndef RTAMD_OBJ_STAY_NUM_MESH
#define RTAMD_OBJ_STAY_NUM_MESH 0
#endif
static constexpr uint32_t OBJ_STAY_NUM_MESH = RTAMD_OBJ_STAY_NUM_MESH;
#ifndef RTAMD_TNODE_STAY_NUM
#define RTAMD_TNODE_STAY_NUM 3
#endif
static constexpr uint32_t TNODE_STAY_NUM = RTAMD_TNODE_STAY_NUM;
#ifndef RTAMD_TNODE_STAY_NUM_MESH
#define RTAMD_TNODE_STAY_NUM_MESH 1
#endif
static constexpr uint32_t TNODE_STAY_NUM_MESH = RTAMD_TNODE_STAY_NUM_MESH;
static constexpr uint32_t NODE_STAY_NUM = RTAMD_NODE_STAY_NUM, NODE_STAY_DEN = RTAMD_NODE_STAY_DEN;
// Vote weights (TRI, NODE, OBJ, FETCH, TNODE): a phase's lane count is multiplied by its weight before the
// comparison, so a phase with a low weight waits until it has gathered proportionally more lanes.  The OBJ and
// FETCH steps are several times longer than a NODE step; with these weights they run once the tree walks of
// the mesh walks of the wave have drained, with every lane that will need them (measured +4 % over equal
// weights; 64 is no better than 16, weighting OBJ over FETCH or TRI over NODE is worse).  Object-tree nodes
// keep weight 1: in a scene of many small objects the OBJ step is the leaf work of that tree, and starving it
// costs a third of the throughput.
#ifndef RTAMD_VOTE_W
#define RTAMD_VOTE_W 16, 16, 1, 1, 1
#endif
// Object-tree walks in passes of growing reach (see tree_span): a ray is limited if it would cross more than DEEP_SPAN
// object spacings; the first pass reaches DEEP_START spacings, each further one DEEP_GROW times as far (*_CHEAP: trees
// whose objects are not meshes).
#ifndef RTAMD_DEEP_START
#define RTAMD_DEEP_START 1
#endif
#ifndef RTAMD_DEEP_SPAN
#define RTAMD_DEEP_SPAN 2
#endif
#ifndef RTAMD_DEEP_START_CHEAP
#define RTAMD_DEEP_START_CHEAP 12
#endif
#ifndef RTAMD_DEEP_SPAN_CHEAP
#define RTAMD_DEEP_SPAN_CHEAP 32
#endif
#ifndef RTAMD_DEEP_GROW
#define RTAMD_DEEP_GROW 4
#endif
#ifndef RTAMD_VOTE_MESH_MIN
#define RTAMD_VOTE_MESH_MIN 16
#endif
static constexpr uint32_t VOTE_MESH_MIN = RTAMD_VOTE_MESH_MIN;
#ifndef RTAMD_NODE_PAIRS
#define RTAMD_NODE_PAIRS 1
#endif
static constexpr bool NODE_PAIRS = RTAMD_NODE_PAIRS != 0;
#ifndef RTAMD_PLAIN_NODE
#define RTAMD_PLAIN_NODE 1
#endif
// mesh-tree boxes of rays that cannot produce a NaN are tested with plain minima and maxima (node_passes_plain)
static constexpr bool PLAIN_NODE = RTAMD_PLAIN_NODE != 0;
#ifndef RTAMD_PLAIN_TNODE
#define RTAMD_PLAIN_TNODE 1
#endif
// the same for the boxes of object trees, chosen per TNODE phase by a wave-uniform flag
static constexpr bool PLAIN_TNODE = RTAMD_PLAIN_TNODE != 0;
#ifndef RTAMD_TNODE_PAIRS
#define RTAMD_TNODE_PAIRS 1
#endif
// object trees of scenes with meshes: two nodes per TNODE step (flattened grid +3.5 %, instanced grid +0.7 %; a tree of spheres
// loses 1.4 %: its walk is bound by instruction issue, and the dropped second tests are instructions)
static constexpr bool TNODE_PAIRS_ON = RTAMD_TNODE_PAIRS != 0;   // mesh trees in global memory: two nodes per NODE step (node_step2)
static constexpr uint32_t VOTE_W[5] = {RTAMD_VOTE_W};
static constexpr uint32_t VOTE_W_OTHER_MAX = (VOTE_W[2] > VOTE_W[3] ? (VOTE_W[2] > VOTE_W[4] ? VOTE_W[2] : VOTE_W[4]) : (VOTE_W[3] > VOTE_W[4] ? VOTE_W[3] : VOTE_W[4]));
static constexpr uint32_t VOTE_W_TRI = VOTE_W[0], VOTE_W_NODE = VOTE_W[1], VOTE_W_OBJ = VOTE_W[2], VOTE_W_FETCH = VOTE_W[3], VOTE_W_TNODE = VOTE_W[4];

// Diagnostic build only (-DRTAMD_DIAG): per-phase cycle, step and active-lane totals of the scheduler,
// accumulated per wave and added to a global table when the wave ends.  Never compiled into librtamd.so.
#ifdef RTAMD_DIAG
// (one copy per kernel file: each file's read-out function reads its own)
// [32..]: frame kernel: waves, sum / max of wave lifetimes, integrator steps, pixels stepped, ticks inside the step, FETCH steps, rays handed
// out, lanes that asked, lanes left without a ray, ticks waiting for the rays' words
static __device__ unsigned long long g_diag[48];
#define DIAG_DECL unsigned long long d_pool[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long d_born = __builtin_amdgcn_s_memtime(); unsigned long long d_cyc[6] = {0, 0, 0, 0, 0, 0}, d_steps[6] = {0, 0, 0, 0, 0, 0}, d_lanes[6] = {0, 0, 0, 0, 0, 0}; unsigned long long d_t0 = 0; unsigned long long d_sub[8] = {0, 0, 0, 0, 0, 0, 0, 0}, d_subn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DIAG_TOP d_t0 = __builtin_amdgcn_s_memtime();
#define DIAG_STEP(ph, c) { d_steps[ph] += 1; d_lanes[ph] += (c); }
#define DIAG_END(ph) { d_cyc[ph] += __builtin_amdgcn_s_memtime() - d_t0; }
#define DIAG_SUB_BEGIN unsigned long long d_s0 = __builtin_amdgcn_s_memtime();
#define DIAG_SUB_END(slot) { d_sub[slot] += __builtin_amdgcn_s_memtime() - d_s0; d_subn[slot] += 1; }
#define DIAG_SUB_MARK d_s0 = __builtin_amdgcn_s_memtime();
#define DIAG_FLUSH if (IO::MEGA && lane == 0) { const unsigned long long life = __builtin_amdgcn_s_memtime() - d_born; atomicAdd(&g_diag[32], 1ull); atomicAdd(&g_diag[33], life); atomicMax(&g_diag[34], life); for (int q = 0; q < 12; q++) atomicAdd(&g_diag[35 + q], d_pool[q]); } if (lane == 0) { for (int q = 0; q < 8; q++) { atomicAdd(&g_diag[15 + 2 * q], d_sub[q]); atomicAdd(&g_diag[16 + 2 * q], d_subn[q]); } } if (lane == 0) { for (int q = 0; q < 5; q++) { atomicAdd(&g_diag[q * 3], d_cyc[q]); atomicAdd(&g_diag[q * 3 + 1], d_steps[q]); atomicAdd(&g_diag[q * 3 + 2], d_lanes[q]); } }
#else
#define DIAG_DECL unsigned long long* const d_pool = nullptr;
#define DIAG_TOP
#define DIAG_STEP(ph, c)
#define DIAG_END(ph)
#define DIAG_SUB_BEGIN
#define DIAG_SUB_END(slot)
#define DIAG_SUB_MARK
#define DIAG_FLUSH
#endif

// Timeline build only (-DRTAMD_TIMELINE): when the scheduler's waves start, run dry and end, without the per-phase probes
// of the diagnostic build (which distort the schedule).  Every wave of the grid has eight words of its own, written with
// plain stores when the wave ends (atomics on shared words would queue up behind the scheduler's own queue-head atomic and
// stretch the very tail they measure): start, moment the queue was found empty (0: never), end of the last launch; summed
// over the launches since the last reset: lifetime, lifetime after the queue ran dry, launches, loop trips.  Stamps are the
// device-wide 100 MHz counter.  The host reduces the table (timeline_read).
#ifdef RTAMD_TIMELINE
static constexpr uint32_t TL_MAX_WAVES = 8192;
__device__ unsigned long long g_timeline[TL_MAX_WAVES * 8];
// the same stamps (start, dry, end, trips) kept per launch: the launch is told apart by its queue-head word (consecutive
// words for the launches of one sample pass), 16 slots
static constexpr uint32_t TL_LAUNCH_SLOTS = 16;
__device__ unsigned long long g_timeline_launch[TL_LAUNCH_SLOTS * TL_MAX_WAVES * 4];
#define TL_DECL unsigned long long tl_start = __builtin_amdgcn_s_memrealtime(), tl_dry = 0, tl_trips = 0;
#define TL_TRIP tl_trips++;
#define TL_DRY if (tl_dry == 0) tl_dry = __builtin_amdgcn_s_memrealtime();
#define TL_FLUSH if (lane == 0 && wave_id < TL_MAX_WAVES) { const unsigned long long tl_end = __builtin_amdgcn_s_memrealtime(); unsigned long long* w = g_timeline + 8 * (size_t)wave_id; \
        w[0] = tl_start; w[1] = tl_dry; w[2] = tl_end; w[3] += tl_end - tl_start; w[4] += tl_dry ? tl_end - tl_dry : 0ull; w[5] += 1ull; w[6] += tl_trips; \
        unsigned long long* q = g_timeline_launch + 4 * ((size_t)(((uintptr_t)work_head >> 2) % TL_LAUNCH_SLOTS) * TL_MAX_WAVES + wave_id); q[0] = tl_start; q[1] = tl_dry; q[2] = tl_end; q[3] = tl_trips; }
#else
#define TL_DECL
#define TL_TRIP
#define TL_DRY
#define TL_FLUSH
#endif

// Attribution build only (-DRTAMD_LDS_DUP=<bits>): the LDS accesses of one kind are issued a second time, into registers that
// nothing uses, so that the growth of SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE over the plain build is that kind's share (results
// and control flow stay what they are).  1 node records, 2 triangle records, 4 staged rays, 8 owner marks of the triangle phase,
// 16 its result slots (atomic minimum + read-back).  Never compiled into librtamd.so.
#ifndef RTAMD_LDS_DUP
#define RTAMD_LDS_DUP 0
#endif
#ifndef RTAMD_TRI_LANE_PERM
#define RTAMD_TRI_LANE_PERM 0
#endif
__device__ __forceinline__ void lds_dup4(lds_float4* p) {
    typedef volatile __attribute__((address_space(3))) const v4f* vp;
    const v4f v = *(vp)p;
    asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
}

// Claims the next chunk of the work queue for the wave (one atomic per chunk).
__device__ __forceinline__ uint32_t claim_chunk(uint32_t* work_head, uint32_t chunk) {
    uint32_t base = 0;
    if (lane_id() == 0) base = atomicAdd(work_head, chunk);
    return __builtin_amdgcn_readfirstlane(base);
}

// Rays a wave takes per batch.  A queue that cannot give every wave of the grid a full batch of 64
// is spread thinner instead: a wave's time on a batch grows with the number of different phase
// steps its rays need, so 4096 waves with 8 rays each finish far sooner than 512 waves with 64.
__device__ __forceinline__ uint32_t sched_batch_size(uint32_t n_items, uint32_t n_waves, uint32_t min_batch) {
    if (n_items >= n_waves * 64u) return 64u;
    const uint32_t bs = (n_items + n_waves - 1u) / n_waves;
    return bs < min_batch ? min_batch : bs;
}
// The scheduler's tuning word: rays per shared claim (bits 0-15), eighths of the queue dealt
// statically (16-19), smallest batch (20-27).
__device__ __forceinline__ uint32_t tune_min_batch(uint32_t tune) { const uint32_t m = (tune >> 20) & 0xFFu; return m < 1u ? 1u : (m > 64u ? 64u : m); }

// Per-wave ray staging buffer in LDS: 8 rows (one per raw word of a ray) of 64 dwords.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
static constexpr uint32_t STAGE_WORDS = 8 * 64;
// One row of the staging buffer: lane l's dword at gsrc lands at lds_dst + 4 * l.  lds_dst is the
// wave-uniform LDS byte address of the row.  M0 carries the LDS base of a direct-to-LDS load and is
// saved and restored inside the statement.
__device__ __forceinline__ void stage_row(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// IO concept:
//   void stage_addr(uint32_t idx, const void* (&src)[8]) const;   // where the 8 raw words of ray idx live
//   void decode(uint32_t idx, const uint32_t (&raw)[8], f3& pos, f3& dir, float& tmax, bool& any, uint32_t& tag) const;
//   void store_closest(uint32_t idx, uint32_t tag, bool found, const Best& b, float t_world) const;
//   void store_any(uint32_t idx, uint32_t tag, bool occluded) const;
//   static constexpr bool MEGA;   // false: rays come from one queue in global memory (work_head, n_items); true: see PixelPool below
// `tag` is an opaque word the IO wants back when the result is stored (loaded with the ray, so that
// storing never needs a dependent read).  Every address stage_addr() returns must be readable, also
// for words the ray kind does not use.
// Per-lane boolean state lives in one VGPR word (`fl`).  Kept as separate bools the compiler holds
// each one as a 64-bit lane mask in SGPRs and spends three scalar instructions per bool at every
// control-flow join of the loop; as bits of a vector register an update is one v_or / v_and.
enum : uint32_t {
    F_ANY = 1u,       // the ray is an occlusion query
    F_HAVE = 2u,      // a finished ray's result waits to be stored
    F_OCC = 4u,       // occlusion query: something was hit
    F_INTREE = 8u,    // walking an object tree
    F_MLDS = 16u,     // the current mesh is the LDS-resident one (MESH_MIXED only)
    F_PEND = 32u,     // a mesh walk ended and has not been folded into the ray's state yet
    F_FOUND = 64u,    // the current mesh walk accepted a triangle
    F_GENXF = 128u,   // the current object's wrapper chain must be walked op by op
    F_ODDW = 256u,    // the ray can turn a slab product of the current object tree into a NaN: its boxes need node_passes
    F_MRANGE = 512u,  // ordered mesh walk: a subtree is being walked in the reference's order (the stack was full)
    F_TORD = 1024u,   // the current object tree is walked nearer child first
    F_TRANGE = 2048u  // ordered object-tree walk: a subtree is being walked in the reference's order
};
// Ordered walks keep the number of postponed children in the flag word: bits 16-19 the mesh walk's, 20-23 the object tree's.
static constexpr uint32_t SPM_SHIFT = 16u, SPT_SHIFT = 20u, SP_MASK = 15u;
// max of two wave-uniform values on the scalar unit
__device__ __forceinline__ uint32_t smax(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("s_max_u32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc");
    return r;
}
__device__ __forceinline__ uint32_t wave_count(bool p) { return (uint32_t)__builtin_popcountll(__ballot(p)); }

// Inclusive wave64 scans on the VALU's lane-crossing paths (no LDS round trips): four row_shr steps
// scan each row of 16 lanes, row_bcast15 / row_bcast31 carry the row totals across.  Must run with
// all 64 lanes enabled; lanes that do not take part contribute the identity (0).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {
    x += dpp_or_zero<0x111, 0xF>(x);
    x += dpp_or_zero<0x112, 0xF>(x);
    x += dpp_or_zero<0x114, 0xF>(x);
    x += dpp_or_zero<0x118, 0xF>(x);
    x += dpp_or_zero<0x142, 0xA>(x);
    x += dpp_or_zero<0x143, 0xC>(x);
    return x;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {
    x = max(x, dpp_or_zero<0x111, 0xF>(x));
    x = max(x, dpp_or_zero<0x112, 0xF>(x));
    x = max(x, dpp_or_zero<0x114, 0xF>(x));
    x = max(x, dpp_or_zero<0x118, 0xF>(x));
    x = max(x, dpp_or_zero<0x142, 0xA>(x));
    x = max(x, dpp_or_zero<0x143, 0xC>(x));
    return x;
}
__device__ __forceinline__ uint32_t lane_pull(uint32_t src_lane, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }
__device__ __forceinline__ float lane_pull(uint32_t src_lane, float v) { return __uint_as_float(lane_pull(src_lane, __float_as_uint(v))); }

// Per-wave LDS scratch of the cooperative triangle phase: owner marks (64 words), the winning key of
// every owner lane (64 x 64 bit) and the winner's t, u, v (3 x 64 words).
static constexpr uint32_t COOP_WORDS = 64 + 128 + 192;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;

// ------------------------------------------------------------------------------------------------
// Walks that visit the nearer child first.  The reference's walk (bvh.rs:227-252) takes the children of a node in the
// order the builder left them; the closest hit it reports does not depend on that order except through the running
// t_max it culls with and through ties: `t <= t_max` lets a later primitive replace an earlier one at the same t, so
// among equal t the last in tree order wins.  A walk in any order that tests every box and primitive with the same
// arithmetic, culls with its own running t_max and resolves equal t by tree order (tri_phase) finds the same hit, because
// the boxes over the nearest hit are entered no later than its t.  One case differs: a triangle the ray is parallel to
// answers t = 0 (triangle.rs:26-28, accepted by mesh.rs:130) wherever along the ray it lies; the reference reports it
// if it reaches the triangle's leaf before a nearer real hit has shrunk t_max below the leaf's box, a nearest-first walk
// usually does not.  Occlusion queries are exact in any order (constant t_max).
//
// Every lane has a short stack of postponed children in LDS: entry e of lane l at words 2 * (64 e + l) of the wave's
// area: {LEAF_BIT | node index, entry parameter t0 of the child's box}.  A popped child whose t0 has fallen behind the
// ray's t_max is dropped without a read.  When both children are hit and the stack is full, the node's whole subtree is
// walked in the reference's order instead (skip links need no stack), then the walk pops on.
//   ni: NI_ROOT the tree's root is to be tested; LEAF_BIT | i: leaf i, taken from the stack, is to be entered; else inner node
//       i, whose two children are tested now (one 64-byte record).  In range mode: the node the reference-order walk
//       stands on, nend its end; that walk tests two consecutive nodes per step, as node_step2 does.
//   Every case reads 64 bytes at one address, so lanes in different cases share the step's one memory round trip.
//   out: leaf (first, end: the caller runs the leaf, then ordered_leaf_done), done (nothing left).
// ------------------------------------------------------------------------------------------------
static constexpr uint32_t NI_ROOT = 0x7FFFFFFFu;
// The next postponed child that still passes `t0 <= t_max + 0.01` (the one part of the visit test that depends on t_max).
// COMPACT: 4-byte entries, for object trees in instances that have little LDS for the stack: {leaf: 1 bit, node index: 17 bits,
// t0 rounded towards minus infinity to sign + exponent + 5 mantissa bits: 14 bits}, entry e of lane l at word 64 e + l.  The
// rounded t0 is never above the exact one, so an entry is never dropped that the exact test would keep; one that is kept within
// the 3 % of the rounding is turned away by the exact tests of the next step.  Trees of up to ORD_COMPACT_MAX_NODES nodes.
static constexpr uint32_t ORD_COMPACT_MAX_NODES = 1u << 17;
__device__ __forceinline__ uint32_t ordered_pack(uint32_t leaf_and_index, float t0) {
    const uint32_t b = __float_as_uint(t0);
    uint32_t q = b >> 18;
    q += ((b >> 31) != 0u && (b & 0x3FFFFu) != 0u) ? 1u : 0u;   // negative: a larger magnitude is further down
    return ((leaf_and_index >> 31) << 31) | ((leaf_and_index & 0x1FFFFu) << 14) | (q & 0x3FFFu);
}
template <bool COMPACT>
__device__ __forceinline__ void ordered_push(lds_u32* slot0, uint32_t sp, uint32_t leaf_and_index, float t0) {
    if (COMPACT) slot0[64u * sp] = ordered_pack(leaf_and_index, t0);
    else *(lds_u64*)(slot0 + 128u * sp) = ((unsigned long long)__float_as_uint(t0) << 32) | leaf_and_index;
}
template <bool COMPACT>
__device__ __forceinline__ bool ordered_pop(uint32_t& ni, uint32_t& sp, lds_u32* slot0, const float tmax) {
    while (sp > 0u) {
        sp--;
        if (COMPACT) {
            const uint32_t e = slot0[64u * sp];
            if (__uint_as_float((e & 0x3FFFu) << 18) <= tmax + 0.01f) { ni = (e & LEAF_BIT) | ((e >> 14) & 0x1FFFFu); return true; }
        } else {
            const unsigned long long e = *(lds_u64*)(slot0 + 128u * sp);
            if (__uint_as_float((uint32_t)(e >> 32)) <= tmax + 0.01f) { ni = (uint32_t)e; return true; }
        }
    }
    return false;
}
template <bool EXACT, bool COMPACT = false>
__device__ __forceinline__ void ordered_step(const float4* nodes, const float4* wide, const f3 pos, const f3 inv, const float tmax, uint32_t& ni, uint32_t& nend,
                                             uint32_t& sp, const uint32_t sp_cap, bool& range, lds_u32* slot0 /* word 0 of this lane's entry 0 */, bool& leaf,
                                             uint32_t& first, uint32_t& end, bool& done) {
    leaf = false; done = false;
    const bool at_root = !range && ni == NI_ROOT;
    const bool popped_leaf = !range && !at_root && (ni & LEAF_BIT) != 0u;
    const uint32_t idx = ni & ~LEAF_BIT;
    const float4* rec = (range || popped_leaf) ? nodes + 2 * (size_t)idx : (at_root ? nodes : wide + 4 * (size_t)idx);
    const float4 a0 = rec[0], a1 = rec[1], b0 = rec[2], b1 = rec[3];
    const uint32_t A1 = __float_as_uint(a1.z), B1 = __float_as_uint(a1.w), A2 = __float_as_uint(b1.z), B2 = __float_as_uint(b1.w);
    const bool leaf1 = (A1 & LEAF_BIT) != 0u, leaf2 = (A2 & LEAF_BIT) != 0u;
    const uint32_t first1 = A1 & ~LEAF_BIT, first2 = A2 & ~LEAF_BIT;
    float t01, t02;
    const bool pass1 = node_test<EXACT>(a0, a1, pos, inv, tmax, t01);
    const bool pass2 = node_test<EXACT>(b0, b1, pos, inv, tmax, t02);
    if (range) {   // nodes ni and ni + 1 of the reference-order walk of [ni, nend)
        const bool enter1 = pass1 && leaf1 && first1 < B1;
        const bool on = !enter1 && (pass1 || leaf1) && ni + 1u < nend;
        const bool enter2 = on && pass2 && leaf2 && first2 < B2;
        const uint32_t next1 = (pass1 || leaf1) ? ni + 1u : A1;
        const uint32_t next2 = enter2 ? ni + 1u : ((pass2 || leaf2) ? ni + 2u : A2);
        ni = enter1 ? ni : (on ? next2 : next1);
        if (enter1 || enter2) { leaf = true; first = enter1 ? first1 : first2; end = enter1 ? B1 : B2; return; }
        if (ni >= nend) { range = false; done = !ordered_pop<COMPACT>(ni, sp, slot0, tmax); }
        return;
    }
    // (its box passed when it was postponed and at the pop; a compact entry's rounded t0 lets a few through that the exact test now turns away)
    if (popped_leaf) {
        if (!COMPACT || pass1) { leaf = true; first = first1; end = B1; }
        else done = !ordered_pop<COMPACT>(ni, sp, slot0, tmax);
        return;
    }
    if (at_root) {
        if (pass1 && leaf1 && first1 < B1) { leaf = true; first = first1; end = B1; return; }
        if (!pass1 || leaf1) { done = true; return; }
        ni = 0u;   // the root's two-child record
        return;
    }
    // a0, a1: the left child (node idx + 1); b0, b1: the right one (the left child's skip link, or idx + 2 behind a left leaf)
    const uint32_t l_idx = idx + 1u, r_idx = leaf1 ? idx + 2u : A1;
    // (an empty leaf is visited and left at once: nothing to do there)
    const bool hit_l = pass1 && (!leaf1 || first1 < B1), hit_r = pass2 && (!leaf2 || first2 < B2);
    const bool both = hit_l && hit_r;
    if (both && sp >= sp_cap) {   // no room to postpone a child: this node's subtree in the reference's order
        range = true; ni = l_idx; nend = leaf2 ? r_idx + 1u : A2;
        return;
    }
    const bool right_first = both ? t02 < t01 : hit_r;   // equal: the left child, as the reference
    if (both) {
        ordered_push<COMPACT>(slot0, sp, right_first ? ((leaf1 ? LEAF_BIT : 0u) | l_idx) : ((leaf2 ? LEAF_BIT : 0u) | r_idx), right_first ? t01 : t02);
        sp++;
    }
    if (hit_l || hit_r) {
        const bool n_leaf = right_first ? leaf2 : leaf1;
        leaf = n_leaf; first = right_first ? first2 : first1; end = right_first ? B2 : B1;
        ni = right_first ? r_idx : l_idx;
    } else {
        done = !ordered_pop<COMPACT>(ni, sp, slot0, tmax);
    }
}
// The leaf handed out by ordered_step is done (tmax: the ray's t_max now).  True: nothing is left of the walk.
template <bool COMPACT = false>
__device__ __forceinline__ bool ordered_leaf_done(uint32_t& ni, const uint32_t nend, uint32_t& sp, bool& range, lds_u32* slot0, const float tmax) {
    if (range) {
        ni++;
        if (ni < nend) return false;
        range = false;
    }
    return !ordered_pop<COMPACT>(ni, sp, slot0, tmax);
}

// The lean form of a scheduler kernel serves scenes that hold nothing but what most scenes hold: meshes and cuboids under
// inline wrapper chains (or, without meshes, spheres and cuboids).  Any other scene runs the general form.
__host__ __device__ constexpr uint32_t sched_features(int meshloc, bool lean) {
    return !lean ? FEAT_ALL : (meshloc == MESH_NONE ? (FEAT_SPHERE | FEAT_CUBOID) : FEAT_CUBOID);
}
// Entries per lane of the ordered walks' stack (sched_stack_depth x 512 bytes per wave), 0: no ordered walks in this instance.
// Mesh trees in global memory: what two 512-thread blocks per CU leave of the 160 KB; scenes without meshes run three blocks per
// CU; beside a mesh staged in LDS three entries fit (rtamd_upload_scene keeps mesh + tables within what is left).  Scenes whose
// tables fit LDS (at most 64 objects) have no tree worth an ordered walk.
// ------------------------------------------------------------------------------------------------
// Per-frame form (IO::MEGA): a wave owns up to 320 pixels for a whole launch and does everything for them.  Its lanes trace
// the rays of those pixels; whenever all rays a pixel waits for have returned, the pixel is resumable, and the wave's
// FETCH step runs the integrator for up to 64 resumable pixels at once (io.step: shade the hit, draw the light samples
// and the scatter direction, start the pixel's next sample), which leaves new rays ready.  Nothing is shared between
// waves, so there is nothing to wait for and no launch boundary between the bounces of a path or the samples of a pixel.
//
// State of pixel p of the wave, one LDS word (io.pix[p]):
// (every lane looks after POOL_PER_LANE = 5 of them in the scans below -- 7 would fit the 2 KB the pool takes over from the ray staging
// buffer, but the scans cost more than the room returns: headline frame 64.6 against 66.2 ms)
//   bits 0-7   rays ready to be traced (bit 0 the closest-hit ray, bit k the k-th shadow ray)
//   bits 8-11  rays handed out or ready whose results have not come back
//   bit  12    the pixel has samples left
//   bits 16-31 index of the sample the pixel works on
// A ray is named by ref = p << 3 | k.  io.pix[POOL_PIXELS .. POOL_PIXELS + 63] is scratch for the lists built here.
// ------------------------------------------------------------------------------------------------
static constexpr uint32_t PW_RDY_MASK = 0xFFu, PW_PEND_ONE = 0x100u, PW_PEND_MASK = 0xF00u, PW_ALIVE = 0x1000u, PW_SAMPLE_SHIFT = 16u;
#ifndef RTAMD_POOL_PER_LANE
#define RTAMD_POOL_PER_LANE 5
#endif
static constexpr uint32_t POOL_PER_LANE = RTAMD_POOL_PER_LANE;   // pixels a lane looks after in the FETCH step's scans (at most 9: the counts' fields)
static constexpr uint32_t POOL_PIXELS = 64u * POOL_PER_LANE, POOL_WORDS = POOL_PIXELS + 64u;
#ifndef RTAMD_SKIP_RUN
#define RTAMD_SKIP_RUN 1
#endif
#ifndef RTAMD_POOL_AHEAD
#define RTAMD_POOL_AHEAD 0
#endif
static constexpr uint32_t POOL_AHEAD = RTAMD_POOL_AHEAD;   // pixels up to this many samples ahead of the wave's last one are served with it
// IO (MEGA) concept, besides decode / store_closest / store_any:
//   lds_u32* pix; uint32_t step_min, trip_limit;   // trip_limit: scheduler steps after which a wave gives up
//   void ray_addr(uint32_t ref, const void* (&src)[8]) const;
//   uint32_t step(uint32_t p, bool active) const;     // all 64 lanes call; returns the pixel's new word
//   void finish(uint32_t lane, uint32_t n_closest, uint32_t n_any) const;
//   void fail(uint32_t lane, uint32_t code) const;
#ifdef RTAMD_DIAG
#define POOL_DIAG(...) __VA_ARGS__
#else
#define POOL_DIAG(...)
#endif
template <class IO, class Start>
__device__ __forceinline__ void pool_fetch(const IO& io, const uint32_t lane, uint32_t& st, uint32_t& fl, uint32_t& ridx, uint32_t& rtag, const Best& best,
                                           const float wtmax, uint32_t& rot, uint32_t& n_closest, uint32_t& n_any, Start&& start,
                                           unsigned long long* dp /* diagnostic build: counters */) {
    volatile lds_u32* pix = io.pix;
    volatile lds_u32* list = io.pix + POOL_PIXELS;
    // (1) the rays that have finished count as returned from here on; their results are written further down, behind the loads of
    // the new rays (a wait for those loads would otherwise wait for the stores before them as well)
    const bool had = st == ST_FETCH && (fl & F_HAVE) != 0u;
    if (had) __hip_atomic_fetch_sub(io.pix + (ridx >> 3), PW_PEND_ONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const bool want = st == ST_FETCH || st == ST_IDLE;
    fl &= want ? ~F_HAVE : ~0u;
    const unsigned long long need = __ballot(want);
    const uint32_t c_want = (uint32_t)__builtin_popcountll(need);
    const uint32_t wrank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
    const uint32_t old_ridx = ridx, old_rtag = rtag;
    bool written = false;   // (wave-uniform)
    auto write_results = [&]() {
        if (!written && had) {
            if (fl & F_ANY) io.store_any(old_ridx, old_rtag, (fl & F_OCC) != 0u);
            else io.store_closest(old_ridx, old_rtag, best.obj != MISS, best, wtmax);
        }
        written = true;
    };
    asm volatile("" ::: "memory");
    // (2) every lane looks at its pixels; the scan order starts at group `rot` (round robin over the FETCH steps).  Pixels on the
    // lowest sample index of the wave (class A) come first in both lists below: a pixel whose paths are long needs several times
    // the steps per sample of one that looks at the sky, and served in turn it would still be on its first samples when the
    // others are done, alone in the wave, one ray at a time.  Served first, the wave's pixels advance through their samples together.
    const uint32_t g = (lane + rot) & 63u;
    constexpr int PPL = (int)POOL_PER_LANE;
    uint32_t w[PPL], own, ownA, incl, inclA, t_rdy, t_res, t_alive, tA_rdy, tA_res, smin;
    auto is_res = [](uint32_t x) { return (x & (PW_ALIVE | PW_PEND_MASK | PW_RDY_MASK)) == PW_ALIVE; };
    auto scan = [&]() {
#pragma unroll
        for (int q = 0; q < PPL; q++) w[q] = pix[POOL_PER_LANE * g + q];
        uint32_t inv = 0u;
#pragma unroll
        for (int q = 0; q < PPL; q++) inv = max(inv, (w[q] & PW_ALIVE) ? 0x10000u - (w[q] >> PW_SAMPLE_SHIFT) : 0u);
        smin = 0x10000u - __builtin_amdgcn_readlane(wave_scan_max(inv), 63);
        own = 0; ownA = 0;
        uint32_t alive_here = 0u;
#pragma unroll
        for (int q = 0; q < PPL; q++) {
            const bool a = (w[q] & PW_ALIVE) && (w[q] >> PW_SAMPLE_SHIFT) <= smin + POOL_AHEAD;
            const uint32_t r = (uint32_t)__builtin_popcount(w[q] & PW_RDY_MASK), e = is_res(w[q]) ? (1u << 13) : 0u;
            own += r + e;
            alive_here |= w[q] & PW_ALIVE;
            ownA += a ? r + e : 0u;
        }
        incl = wave_scan_add(own);
        inclA = wave_scan_add(ownA);
        const uint32_t tot = __builtin_amdgcn_readlane(incl, 63), totA = __builtin_amdgcn_readlane(inclA, 63);
        // (counts of ready rays in bits 0-12, of resumable pixels above them: at most 8 x 64 x 9 and 64 x 9)
        t_rdy = tot & 0x1FFFu; t_res = tot >> 13; t_alive = __ballot(alive_here != 0u) != 0ull ? 1u : 0u;
        tA_rdy = totA & 0x1FFFu; tA_res = totA >> 13;
    };
    POOL_DIAG(unsigned long long pt = __builtin_amdgcn_s_memtime();)
    scan();
    POOL_DIAG({ const unsigned long long n = __builtin_amdgcn_s_memtime(); dp[7] += n - pt; pt = n; })
    // (3) resumable pixels: run the integrator for them when there are enough to fill the wave, or when lanes would
    // otherwise go without a ray
    if (t_res > 0u && (t_res >= io.step_min || t_rdy < c_want)) {
        uint32_t posA = (inclA - ownA) >> 13;
        uint32_t posB = tA_res + ((incl - own) >> 13) - posA;
#pragma unroll
        for (int q = 0; q < PPL; q++)
            if (is_res(w[q])) {
                const bool a = (w[q] >> PW_SAMPLE_SHIFT) <= smin + POOL_AHEAD;
                const uint32_t pos = a ? posA++ : posB++;
                if (pos < 64u) list[pos] = POOL_PER_LANE * g + q;
            }
        const uint32_t n_step = t_res < 64u ? t_res : 64u;
        const bool act = lane < n_step;
        const uint32_t p = act ? list[lane] : 0u;
        write_results();   // (the step reads them)
        POOL_DIAG(const unsigned long long ps0 = __builtin_amdgcn_s_memtime();)
        const uint32_t nw = io.step(p, act);
        POOL_DIAG(dp[0] += 1; dp[1] += n_step; dp[2] += __builtin_amdgcn_s_memtime() - ps0;)
        if (act) pix[p] = nw;
        asm volatile("" ::: "memory");
        scan();
        POOL_DIAG({ const unsigned long long n = __builtin_amdgcn_s_memtime(); dp[8] += n - pt; pt = n; })
    }
    // (4) ready rays to the lanes that want one
    const uint32_t take = c_want < t_rdy ? c_want : t_rdy;
    if (take > 0u) {
        uint32_t posA = (inclA - ownA) & 0x1FFFu;
        uint32_t posB = tA_rdy + ((incl - own) & 0x1FFFu) - posA;
        bool any_handed = false;
#pragma unroll
        for (int q = 0; q < PPL; q++) {
            uint32_t bits = w[q] & PW_RDY_MASK, handed = 0u;
            const bool a = (w[q] >> PW_SAMPLE_SHIFT) <= smin + POOL_AHEAD;
            uint32_t pos = a ? posA : posB;
            while (bits != 0u && pos < take) {
                const uint32_t k = (uint32_t)__builtin_ctz(bits);
                list[pos++] = ((POOL_PER_LANE * g + q) << 3) | k;
                handed |= 1u << k;
                bits &= bits - 1u;
            }
            pos += (uint32_t)__builtin_popcount(bits);
            if (a) posA = pos; else posB = pos;
            if (handed) { pix[POOL_PER_LANE * g + q] = w[q] & ~handed; any_handed = true; }
        }
        // the next step starts with the group this one ended in
        const unsigned long long served = __ballot(any_handed);
        rot = (rot + 63u - (uint32_t)__builtin_clzll(served)) & 63u;
    }
    POOL_DIAG({ const unsigned long long n = __builtin_amdgcn_s_memtime(); dp[9] += n - pt; pt = n; })
    const bool got = want && wrank < take;
    uint32_t raw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (got) {
        ridx = list[wrank];
        const void* src[8];
        io.ray_addr(ridx, src);
#pragma unroll
        for (int c = 0; c < 8; c++) raw[c] = *(const uint32_t*)src[c];
    }
    write_results();
    if (got) start(raw);
    else if (want) {
        // nothing for this lane: it parks until the next FETCH step (which a lane that finishes a ray brings about); while
        // resumable pixels are left it keeps asking, so that the step for them comes even if no ray is in flight
        st = t_alive == 0u ? ST_DONE : (t_res > 0u ? ST_FETCH : ST_IDLE);
    }
    POOL_DIAG({ asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long n = __builtin_amdgcn_s_memtime(); dp[10] += n - pt; })
    POOL_DIAG(dp[3] += 1; dp[4] += take; dp[5] += c_want; dp[6] += c_want - take;)
    const unsigned long long got_any = __ballot(got && (ridx & 7u) != 0u);
    n_any += (uint32_t)__builtin_popcountll(got_any);
    n_closest += take - (uint32_t)__builtin_popcountll(got_any);
}

__host__ __device__ constexpr uint32_t sched_stack_depth(int meshloc, bool small_scene) {
    return meshloc == MESH_GLOBAL ? 10u : (small_scene ? 0u : (meshloc == MESH_NONE ? 7u : 3u));
}
template <int MESHLOC, bool LEAN, class Tab, class IO>
__device__ __forceinline__ void trace_scheduler(const DevScene& S, const Tab& tab, const LdsMesh lds, const IO& io, uint32_t* work_head,
                                                const uint32_t n_items, const uint32_t chunk_and_share, lds_u32* stage /* this wave's STAGE_WORDS */,
                                                lds_u32* coop /* this wave's COOP_WORDS */, lds_u32* ostack /* this wave's 128 * sched_stack_depth words */) {
    constexpr uint32_t SD = sched_stack_depth(MESHLOC, Tab::small_scene);
    // mesh trees in global memory / object trees of scenes whose tables do not fit LDS can be walked nearer child first
    constexpr bool ORD_M = SD > 0u && MESHLOC == MESH_GLOBAL, ORD_T = SD > 0u && !Tab::small_scene;
    const bool ord_m = ORD_M && (S.ordered & ORD_MESH) != 0u, ord_t = ORD_T && (S.ordered & ORD_TREE) != 0u;   // wave-uniform
    // the object tree leaves two entries to the mesh walks inside it
    // (alone in the area -- no mesh walks on the stack -- the tree's entries are the compact ones: twice as many)
    constexpr bool TC = !ORD_M;
    constexpr uint32_t SD_T = ORD_M ? (SD > 2u ? SD - 2u : 0u) : (2u * SD < 15u ? 2u * SD : 15u);
    const uint32_t cap_m = SD < S.ordered_cap ? SD : S.ordered_cap, cap_t = SD_T < S.ordered_cap ? SD_T : S.ordered_cap;   // wave-uniform
    const uint32_t chunk = chunk_and_share & 0xFFFFu, share8 = (chunk_and_share >> 16) & 0xFu;
    const uint32_t lane = lane_id();
    constexpr uint32_t FEAT = sched_features(MESHLOC, LEAN);
    constexpr bool GENERIC_XF = (FEAT & FEAT_GENERIC_XF) != 0u;

    uint32_t st = ST_FETCH, fl = 0;
    uint32_t ridx = 0, rtag = 0;
    f3 wpos = mk(0, 0, 0), wdir = mk(0, 0, 1), winv = mk(0, 0, 0);
    float wtmax = 0.0f;
    // staged batch: row c of `stage` holds raw word c of rays sg_base .. sg_base + sg_cnt - 1; sg_pos are handed out
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)stage);
    uint32_t sg_base = 0, sg_cnt = 0, sg_pos = 0;  // wave-uniform
    Best best;
    best.obj = MISS; best.prim = 0; best.t_local = 0.f; best.u = 0.f; best.v = 0.f;
    uint32_t root_i = 0;
    uint32_t t_i = 0, t_n = 0, t_base = 0, t_first = 0;
    float t_lim = __builtin_inff();   // object-tree walk: nothing beyond this ray parameter is visited in the current pass (see tree_span)
    uint32_t o_cur = 0, o_end = 0;
    uint32_t oi = 0;
    f3 opos = mk(0, 0, 0), odir = mk(0, 0, 1), oinv = mk(0, 0, 0);
    float otmax = 0.0f, oscale = 1.0f;
    uint32_t n_i = 0, n_n = 0, node_off = 0, tri_off = 0;
    uint32_t k_cur = 0, k_end = 0, oprim = 0;
    float ou = 0.f, ov = 0.f;
    // Work distribution (all wave-uniform).  The first part of the queue is dealt out statically:
    // wave w owns the 64-ray batches w, w + W, w + 2W, ... (W waves in the grid), spread over the
    // whole queue so that every wave sees a mix of cheap and expensive rays, and needs no atomic
    // (batches are shorter than 64 rays when the queue is short, see sched_batch_size).
    // The rest is claimed `chunk` rays at a time from a shared head, which evens
    // out what the static deal left unbalanced.  A returning atomic on one address completes at
    // under 100 per microsecond on this GPU, so a launch must not start with one per wave.
    // wave numbering runs over the blocks first, so that a short queue lands on every CU
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * gridDim.x + blockIdx.x);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t bs = sched_batch_size(n_items, n_waves, tune_min_batch(chunk_and_share));   // 64 unless the queue is short
    const uint32_t n_batches = (n_items + bs - 1u) / bs;
    uint32_t k_static = n_batches <= n_waves ? 1u : (share8 * n_batches) / (8u * n_waves);
    if (k_static == 0u) k_static = 1u;
    // A queue of between one and two batches per wave: what the first batches leave is dealt too, as one shorter batch per
    // wave.  Left to the shared head it would go to the waves whose claims arrive first -- every wave asks as soon as its
    // first batch is handed out -- and those would hold twice the rays of the others, a launch of two batch times
    // (measured on 0.39 / 0.52 M incoherent rays: 127 / 158 us shared, 108 us for 0.52 M dealt whole).
    uint32_t tail_bs = 0;
    if ((chunk_and_share >> 28) & 1u) {
        const uint64_t full = (uint64_t)n_waves * 64u;
        if (k_static == 1u && n_items > full && n_items <= 2u * full) tail_bs = (uint32_t)((n_items - full + n_waves - 1u) / n_waves);
        else if (k_static == 1u && n_items > 2u * full) k_static = 2u;   // (the same rush for the shared head otherwise; by the second batch the waves are apart)
    }
    const uint64_t dyn_base64 = (uint64_t)k_static * n_waves * bs;
    const uint32_t k_rounds = k_static + (tail_bs != 0u ? 1u : 0u);
    const uint32_t dyn_base = (tail_bs == 0u && dyn_base64 < n_items) ? (uint32_t)dyn_base64 : n_items;
    uint32_t ks = 0;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;                   // nothing is left for this wave to claim
    uint32_t pool_rot = 0, n_closest_taken = 0, n_any_taken = 0;   // (IO::MEGA; wave-uniform)

    // Object trees are walked in the reference's fixed order, which is not front to back: a ray that grazes a scene of
    // thousands of objects visits every object whose box it crosses until the walk happens upon its nearest hit, and a
    // launch cannot end before its slowest ray.  The walk is therefore repeated with a growing limit on the ray parameter:
    // a pass visits what the reference's walk would visit had it started with t_max = limit.  Such a pass is exact once it
    // accepts a hit: box tests are monotone in t_max (`t0 <= t_max + 0.01`), so a pass with a smaller t_max visits a subset
    // of the reference's nodes and accepts a subset of its hits, and every hit the reference accepts at t <= limit is still
    // reached, because the boxes above it are entered no later than t plus rounding, which stays below the 0.01 slack
    // while the ray parameters involved are below 2048 (checked per ray; other rays walk once, unlimited).  From then on
    // the running t_max of the pass equals the reference's.  A pass that accepts nothing leaves the ray untouched and
    // the next one starts from the root with four times the reach; past the tree's bounds the limit is dropped.
    // Scenes whose tables fit LDS hold at most 64 objects: no tree of theirs is worth it, and the code is compiled out.
    constexpr bool DEEPEN = !Tab::small_scene;
    // (entry and exit parameter of the ray through the root box of the current object tree)
    auto tree_span = [&](float& t_en, float& t_ex) -> float {
        const float4* np = S.nodes + 2 * (size_t)t_base;
        const float4 n0 = np[0], n1 = np[1];
        const f3 pa = (mk(n0.x, n0.y, n0.z) - wpos) * winv, pb = (mk(n0.w, n1.x, n1.y) - wpos) * winv;
        t_en = max2(min2(pa.x, pb.x), max2(min2(pa.y, pb.y), min2(pa.z, pb.z)));
        t_ex = min2(max2(pa.x, pb.x), min2(max2(pa.y, pb.y), max2(pa.z, pb.z)));
        // spacing of the objects, taken as spread over the two largest extents of the box (a guess that only steers the
        // limits: any value gives the same hits)
        const float ex = n0.w - n0.x, ey = n1.x - n0.y, ez = n1.y - n0.z;
        const float lo = min2(ex, min2(ey, ez));
        const float area = lo == ex ? ey * ez : (lo == ey ? ex * ez : ex * ey);
        return sqrtf(area / (0.5f * (float)t_n));
    };
    // Only a ray that would cross many objects is limited (the passes of a short walk cost more than they save): the
    // first pass reaches 8 object spacings into the tree's box, each further one four times as far.
    auto walk_begin = [&](const uint32_t tree_deepen) {
        t_lim = __builtin_inff();
        if (DEEPEN && S.deepen && (tree_deepen & TREE_DEEPEN_MASK)) {
            float t_en, t_ex;
            const float cell = tree_span(t_en, t_ex);
            const float s0 = t_en > 0.0f ? t_en : 0.0f;
            // Trees of mesh objects (tree_deepen = 1): every object a pass leaves out is a whole mesh walk.  Trees of cheap
            // objects (spheres, cuboids; 2): an object costs less than a second descent from the root, only the long rays
            // are limited, and later.
            const bool of_meshes = (tree_deepen & TREE_DEEPEN_MASK) == 1u;
            const float start = s0 + (of_meshes ? (float)RTAMD_DEEP_START : (float)RTAMD_DEEP_START_CHEAP) * cell;
            const float span = (of_meshes ? (float)RTAMD_DEEP_SPAN : (float)RTAMD_DEEP_SPAN_CHEAP) * cell;
            // (every comparison is false on NaN: such rays walk once, unlimited)
            // (a limit that does not lie beyond the entry point -- a degenerate root box, or a spacing below the resolution of
            // the entry parameter -- could never grow: such a walk is not limited)
            if (fabsf(t_en) < 2048.0f && fabsf(t_ex) < 2048.0f && t_ex - s0 > span && start < wtmax && start > s0) t_lim = start;
        }
    };
    // The walk has ended.  True: it found nothing within its limit and starts over with a wider one.
    auto walk_again = [&]() -> bool {
        if (!DEEPEN || !(t_lim < wtmax)) return false;
        float t_en, t_ex;
        (void)tree_span(t_en, t_ex);
        const float s0 = t_en > 0.0f ? t_en : 0.0f;
        const float wider = s0 + (t_lim - s0) * (float)RTAMD_DEEP_GROW;
        t_lim = (wider > t_lim && wider < t_ex + 0.02f) ? wider : __builtin_inff();   // (a limit that cannot grow is dropped)
        t_i = 0;
        return true;
    };
    // t_max of the object-tree walk and of the objects entered from it
    auto walk_tmax = [&]() -> float { return (DEEPEN && t_lim < wtmax) ? t_lim : wtmax; };

    // One node of the current mesh tree (bvh.rs:227-252): visit test, then into the leaf, on to the next
    // node, or along the skip link.  Used by the NODE phase and, for a mesh's root, by the OBJ phase.
    // Wave-uniform: some lane inside an object tree needs the reference's NaN rules for the tree's boxes (or a tree holds a box
    // that is not plain).  Set after the OBJ phase, where lanes enter trees, and left standing until that phase runs again.
    // (not in the general form of the mesh-free kernel, which has no registers to spare for a second copy of the phase, nor for
    // scenes whose tables fit LDS: at most 64 objects, whose trees, if any, are walked in a few steps)
    constexpr bool PLAIN_T = PLAIN_TNODE && !Tab::small_scene && (LEAN || MESHLOC != MESH_NONE);
    bool tree_exact = !PLAIN_T || S.plain_tree_boxes == 0u;
    // (PLAIN_NODE: a ray whose slab products could be NaNs never gets here; its mesh walk is done where it enters the mesh)
    auto node_step = [&]() {
        float4 n0, n1;
        if (MESHLOC == MESH_LDS || (MESHLOC == MESH_MIXED && (fl & F_MLDS))) {
            lds_float4* np = lds.nodes + 2 * n_i;
            n0 = lds_load4(np); n1 = lds_load4(np + 1);
            if (RTAMD_LDS_DUP & 1) { lds_dup4(np); lds_dup4(np + 1); }
        } else {
            const float4* np = S.nodes + 2 * ((size_t)node_off + n_i);
            n0 = np[0]; n1 = np[1];
        }
        const uint32_t A = __float_as_uint(n1.z), B = __float_as_uint(n1.w);
        const bool leaf = (A & LEAF_BIT) != 0u;
        const uint32_t first = A & ~LEAF_BIT;
        const bool pass = PLAIN_NODE ? node_passes_plain(n0, n1, opos, oinv, otmax)
                                     : node_passes(n0, n1, opos, oinv, signbit32(oinv.x), signbit32(oinv.y), signbit32(oinv.z), otmax);
        const bool enter = pass && leaf && first < B;
        k_cur = enter ? first : k_cur; k_end = enter ? B : k_end;
        n_i = enter ? n_i : ((pass || leaf) ? n_i + 1 : A);   // skip link on a failed inner node
        const bool mesh_done = !enter && n_i >= n_n;
        fl |= mesh_done ? F_PEND : 0u;
        st = enter ? ST_TRI : (mesh_done ? ST_OBJ : ST_NODE);
    };

    // Two nodes of the current mesh tree in one step, for scenes with trees in global memory: the node the walk stands on and the one
    // stored behind it, which is where the walk goes next unless it enters a leaf or follows a skip link (a passed inner
    // node, or a leaf that is missed or empty).  Both records are requested together (one L2 / HBM round trip, often one
    // cache line) and tested against the same t_max: nothing between the two visits changes it, so the second test is the
    // one the walk would make, and it is dropped when the walk does not go there.  Same visits in the same order as
    // node_step twice.  (For a tree in LDS the second test costs what the saved step returns.)
    auto node_step2 = [&]() {
        const uint32_t i2 = n_i + 1u < n_n ? n_i + 1u : n_i;   // (the last node has nothing behind it)
        float4 n0, n1, m0, m1;
        if (MESHLOC == MESH_MIXED && (fl & F_MLDS)) {   // (a mixed scene's lanes on the LDS-resident mesh take the same step)
            lds_float4* np = lds.nodes + 2 * n_i;
            lds_float4* mp = lds.nodes + 2 * i2;
            n0 = lds_load4(np); n1 = lds_load4(np + 1); m0 = lds_load4(mp); m1 = lds_load4(mp + 1);
        } else {
            const float4* np = S.nodes + 2 * ((size_t)node_off + n_i);
            const float4* mp = S.nodes + 2 * ((size_t)node_off + i2);
            n0 = np[0]; n1 = np[1]; m0 = mp[0]; m1 = mp[1];
        }
        const bool sx = signbit32(oinv.x), sy = signbit32(oinv.y), sz = signbit32(oinv.z);
        const uint32_t A1 = __float_as_uint(n1.z), B1 = __float_as_uint(n1.w), A2 = __float_as_uint(m1.z), B2 = __float_as_uint(m1.w);
        const bool leaf1 = (A1 & LEAF_BIT) != 0u, leaf2 = (A2 & LEAF_BIT) != 0u;
        const uint32_t first1 = A1 & ~LEAF_BIT, first2 = A2 & ~LEAF_BIT;
        const bool pass1 = PLAIN_NODE ? node_passes_plain(n0, n1, opos, oinv, otmax) : node_passes(n0, n1, opos, oinv, sx, sy, sz, otmax);
        const bool pass2 = PLAIN_NODE ? node_passes_plain(m0, m1, opos, oinv, otmax) : node_passes(m0, m1, opos, oinv, sx, sy, sz, otmax);
        const bool enter1 = pass1 && leaf1 && first1 < B1;
        const bool on = !enter1 && (pass1 || leaf1) && n_i + 1u < n_n;   // the walk goes on to the node behind
        const bool enter2 = on && pass2 && leaf2 && first2 < B2;
        const uint32_t next1 = (pass1 || leaf1) ? n_i + 1u : A1;        // skip link on a failed inner node
        const uint32_t next2 = enter2 ? n_i + 1u : ((pass2 || leaf2) ? n_i + 2u : A2);
        const bool enter = enter1 || enter2;
        k_cur = enter1 ? first1 : (enter2 ? first2 : k_cur);
        k_end = enter1 ? B1 : (enter2 ? B2 : k_end);
        n_i = enter1 ? n_i : (on ? next2 : next1);
        const bool mesh_done = !enter && n_i >= n_n;
        fl |= mesh_done ? F_PEND : 0u;
        st = enter ? ST_TRI : (mesh_done ? ST_OBJ : ST_NODE);
    };

    // One step of the ordered walk of the current mesh tree (ordered_step): n_n doubles as the end of a subtree walked in
    // the reference's order.
    auto node_step_ord = [&]() {
        uint32_t spm = (fl >> SPM_SHIFT) & SP_MASK;
        const uint32_t spt = ORD_T ? (fl >> SPT_SHIFT) & SP_MASK : 0u;
        bool range = (fl & F_MRANGE) != 0u, leaf, done;
        uint32_t first = 0u, end = 0u;
        ordered_step<!PLAIN_NODE>(S.nodes + 2 * (size_t)node_off, S.wide + 4 * (size_t)node_off, opos, oinv, otmax, n_i, n_n, spm, cap_m > spt ? cap_m - spt : 0u, range,
                                  ostack + 2u * lane + 128u * spt, leaf, first, end, done);
        fl = (fl & ~(F_MRANGE | (SP_MASK << SPM_SHIFT))) | (range ? F_MRANGE : 0u) | (spm << SPM_SHIFT) | (done ? F_PEND : 0u);
        k_cur = leaf ? first : k_cur; k_end = leaf ? end : k_end;
        st = leaf ? ST_TRI : (done ? ST_OBJ : ST_NODE);
    };

    DIAG_DECL
    TL_DECL
    auto tri_phase = [&]() {
          // Cooperative triangle phase.  A leaf of the reference's trees holds many triangles (7-30 in
          // the bunny), and the lanes waiting here are a minority of the wave, so one lane per ray
          // would test them one by one with most of the wave idle.  Instead the pending (ray, triangle)
          // pairs of all waiting lanes are laid out in one list and every lane of the wave tests one
          // pair per round.  The reference walks a leaf in order, replacing the hit whenever
          // t <= t_max (mesh.rs:130): the outcome is the smallest accepted t, the LAST triangle among
          // equal t.  The same outcome comes from testing every pair against the t_max the ray arrived
          // with and taking the minimum of (t, -index) per ray, done with a 64-bit LDS atomic min.
          // volatile: the lanes talk to each other through these arrays; to the compiler every lane is a
          // thread of its own, and it would otherwise forward a lane's own store to its later load
          volatile lds_u32* mark = coop;
          volatile lds_u64* slot = (volatile lds_u64*)(coop + 64);
          volatile lds_u32* pay = coop + 192;
          {
            // every pair pending at this visit is tested against the t_max its ray arrived with; leaves that span
            // rounds keep accumulating their minimum in the same LDS slot
            const bool is_tri = st == ST_TRI;
            const uint32_t cnt = is_tri ? k_end - k_cur : 0u;
            const uint32_t incl = wave_scan_add(cnt);
            const uint32_t off = incl - cnt;                                   // first pair of this lane's leaf
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            const uint32_t kbase = k_cur - off;                                // triangle of pair p: kbase + p
            slot[lane] = ~0ull;
            // One round = 64 pairs, one per lane.  Who owns pair p = base + lane: owners mark their first pair of the round,
            // a max-scan spreads the mark.
            auto owner_of = [&](const uint32_t base) -> uint32_t {
                mark[lane] = 0u;
                const uint32_t lo = off > base ? off : base;
                if (is_tri && lo < off + cnt && lo < base + 64u) mark[lo - base] = lane + 1u;
                if (RTAMD_LDS_DUP & 8) {
                    if (is_tri && lo < off + cnt && lo < base + 64u) mark[lo - base] = lane + 1u;
                    const uint32_t z = mark[lane];
                    asm volatile("" ::"v"(z));
                }
                return wave_scan_max(mark[lane]);
            };
            struct Pair { uint32_t o, k, tri_off; bool valid, in_lds; f3 pos, dir; float tmax; };
            // the owner's ray and leaf position, pulled over to the lane that tests the pair
            auto pull_pair = [&](const uint32_t base, const uint32_t owner_in_order) -> Pair {
                Pair P;
                // Which pair of the round a lane tests.  A 16-byte LDS read is served in four groups of sixteen lanes that are not
                // consecutive lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same in the upper half); only lanes of one
                // group can collide on a bank, and consecutive 48-byte triangle records never do.  With pair = lane a group's
                // sixteen lanes read pieces of up to three leaves; with the pairs dealt so that a group reads sixteen consecutive
                // pairs it is one leaf, or the end of one and the start of the next.
                uint32_t pl = lane, owner1 = owner_in_order;
                if (RTAMD_TRI_LANE_PERM && MESHLOC != MESH_GLOBAL) {
                    const uint32_t l5 = lane & 31u;
                    const bool g1 = (l5 >= 4u && l5 < 12u) || (l5 >= 16u && l5 < 20u) || l5 >= 28u;
                    const uint32_t pos = l5 < 4u ? l5 : (l5 < 12u ? l5 - 4u : (l5 < 20u ? l5 - 8u : (l5 < 28u ? l5 - 12u : l5 - 16u)));
                    pl = (lane & 32u) + (g1 ? 16u : 0u) + pos;
                    owner1 = lane_pull(pl, owner_in_order);
                }
                const uint32_t p = base + pl;
                P.valid = p < total;
                P.o = P.valid ? owner1 - 1u : 0u;
                const uint32_t k = lane_pull(P.o, kbase) + p;
                P.k = P.valid ? k : 0u;     // lanes without a pair read triangle 0 of a mesh and drop the result
                P.pos = mk(lane_pull(P.o, opos.x), lane_pull(P.o, opos.y), lane_pull(P.o, opos.z));
                P.dir = mk(lane_pull(P.o, odir.x), lane_pull(P.o, odir.y), lane_pull(P.o, odir.z));
                P.tmax = lane_pull(P.o, otmax);
                P.in_lds = MESHLOC == MESH_LDS;
                P.tri_off = 0;
                if (MESHLOC != MESH_LDS) {
                    P.tri_off = lane_pull(P.o, tri_off);
                    if (MESHLOC == MESH_MIXED) P.in_lds = (lane_pull(P.o, fl) & F_MLDS) != 0u;
                }
                return P;
            };
            auto load_tri = [&](const Pair& P, float4& q0, float4& q1, float4& q2) {
                if (P.in_lds) {
                    q0 = lds_load4(lds.tris + P.k); q1 = lds_load4(lds.tris1 + P.k);
                    q2 = make_float4(lds.trisz[P.k], 0.f, 0.f, 0.f);
                    if (RTAMD_LDS_DUP & 2) {
                        lds_dup4(lds.tris + P.k); lds_dup4(lds.tris1 + P.k);
                        const float z = ((volatile __attribute__((address_space(3))) const float*)lds.trisz)[P.k];
                        asm volatile("" ::"v"(z));
                    }
                } else {
                    const float4* tp = S.tris + 3 * ((size_t)P.tri_off + P.k);
                    q0 = tp[0]; q1 = tp[1]; q2 = tp[2];
                }
            };
            auto commit = [&](const Pair& P, const bool hit, const float t, const float u, const float v) {
                // accepted t is >= 0 (or -0): its bit pattern orders like its value once -0 is folded onto +0
                const uint32_t t_key = (t == 0.0f) ? 0u : __float_as_uint(t);
                const unsigned long long my_key = ((unsigned long long)t_key << 32) | (unsigned long long)(0xFFFFFFFFu - P.k);
                if (hit) {
                    __hip_atomic_fetch_min((lds_u64*)(coop + 64) + P.o, my_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (RTAMD_LDS_DUP & 16) {
                        __hip_atomic_fetch_min((lds_u64*)(coop + 64) + P.o, my_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const unsigned long long z = slot[P.o];
                        asm volatile("" ::"v"((uint32_t)z), "v"((uint32_t)(z >> 32)));
                    }
                    if (slot[P.o] == my_key) { pay[P.o] = __float_as_uint(t); pay[64 + P.o] = __float_as_uint(u); pay[128 + P.o] = __float_as_uint(v); }
                }
            };
            // Two rounds at a time while more than 64 pairs are left: their tests are independent instruction streams,
            // which the SIMD issues at nearly twice the rate of one dependent chain.
            uint32_t base = 0;
            for (; base + 64u < total; base += 128u) {
                DIAG_STEP(ST_TRI, 64u)
                DIAG_STEP(ST_TRI, (total - base - 64u < 64u ? total - base - 64u : 64u))
                const uint32_t own0 = owner_of(base);
                const uint32_t own1 = owner_of(base + 64u);
                const Pair P0 = pull_pair(base, own0), P1 = pull_pair(base + 64u, own1);
                float4 a0, a1, a2, b0, b1, b2;
                load_tri(P0, a0, a1, a2);
                load_tri(P1, b0, b1, b2);
                float t0, u0, v0, t1, u1, v1;
                const bool h0 = tri_test_flat(a0, a1, a2, P0.pos, P0.dir, P0.tmax, t0, u0, v0) & P0.valid;
                const bool h1 = tri_test_flat(b0, b1, b2, P1.pos, P1.dir, P1.tmax, t1, u1, v1) & P1.valid;
                // both tests are complete before the first result is committed (the commit is control flow, which the
                // instruction scheduler does not move code across)
                asm volatile("" : "+v"(t0), "+v"(u0), "+v"(v0), "+v"(t1), "+v"(u1), "+v"(v1));
                commit(P0, h0, t0, u0, v0);
                commit(P1, h1, t1, u1, v1);
            }
            if (base < total) {
                DIAG_STEP(ST_TRI, (total - base < 64u ? total - base : 64u))
                const Pair P0 = pull_pair(base, owner_of(base));
                float4 a0, a1, a2;
                load_tri(P0, a0, a1, a2);
                float t0, u0, v0;
                const bool h0 = tri_test_flat(a0, a1, a2, P0.pos, P0.dir, P0.tmax, t0, u0, v0) & P0.valid;
                commit(P0, h0, t0, u0, v0);
            }
            // owners: take the result; the leaf is done: i += 1 (bvh.rs:245)
            if (is_tri) {
                const unsigned long long res = slot[lane];
                const bool found = res != ~0ull;
                const bool is_any = (fl & F_ANY) != 0u;
                const bool hit_a = found && is_any;
                bool hit_c = found && !is_any;
                // (an ordered walk can come upon a triangle at the t of the hit it holds after a later one in tree order: among
                // equal t the reference keeps the last in tree order, mesh.rs:130)
                if (ORD_M && ord_m && hit_c && (fl & F_FOUND) && __uint_as_float(pay[lane]) == otmax && 0xFFFFFFFFu - (uint32_t)res < oprim) hit_c = false;
                if (hit_c) {
                    otmax = __uint_as_float(pay[lane]); ou = __uint_as_float(pay[64 + lane]); ov = __uint_as_float(pay[128 + lane]);
                    oprim = 0xFFFFFFFFu - (uint32_t)res;
                }
                k_cur = k_end;
                bool mesh_done;
                if (ORD_M && ord_m) {
                    bool range = (fl & F_MRANGE) != 0u;
                    uint32_t spm = (fl >> SPM_SHIFT) & SP_MASK;
                    const uint32_t spt = ORD_T ? (fl >> SPT_SHIFT) & SP_MASK : 0u;
                    mesh_done = hit_a || ordered_leaf_done(n_i, n_n, spm, range, ostack + 2u * lane + 128u * spt, otmax);
                    fl = (fl & ~(F_MRANGE | (SP_MASK << SPM_SHIFT))) | (range ? F_MRANGE : 0u) | (spm << SPM_SHIFT);
                } else {
                    n_i++;
                    mesh_done = n_i >= n_n;
                }
                fl |= (hit_c ? F_FOUND : 0u) | (hit_a ? (F_OCC | F_HAVE) : 0u) | ((mesh_done && !hit_a) ? F_PEND : 0u);
                st = hit_a ? ST_FETCH : (mesh_done ? ST_OBJ : ST_NODE);
            }
          }
    };
    auto node_phase = [&](const uint32_t c_first) {
          for (uint32_t c = c_first; c * NODE_STAY_DEN >= c_first * NODE_STAY_NUM && (NODE_STAY_NUM > 0u || c > 0u); c = wave_count(st == ST_NODE)) {
            DIAG_STEP(ST_NODE, c)
            if (st == ST_NODE) {
                if (ORD_M && ord_m) node_step_ord();
                else if (NODE_PAIRS && (MESHLOC == MESH_GLOBAL || MESHLOC == MESH_MIXED)) node_step2();
                else node_step();
            }
          }
    };
    uint32_t pool_trips = 0;
    for (;;) {
        DIAG_TOP
        TL_TRIP
        if constexpr (IO::MEGA) {   // a wave that does not come to an end gives up (and says so) rather than hold the GPU
            if (++pool_trips > io.trip_limit) { io.fail(lane, 1u); break; }
        }
        // ---- vote: run the phase most lanes are waiting for ----
        const uint32_t c_tri = MESHLOC != MESH_NONE ? wave_count(st == ST_TRI) : 0u;
        const uint32_t c_node = MESHLOC != MESH_NONE ? wave_count(st == ST_NODE) : 0u;
        const uint32_t c_tnode = wave_count(st == ST_TNODE);
        const uint32_t c_obj = wave_count(st == ST_OBJ);
        const uint32_t c_fetch = wave_count(st == ST_FETCH);
        if ((c_tri | c_node | c_tnode | c_obj | c_fetch) == 0) break;  // every lane is ST_DONE
        // most lanes wins; ties go to the deeper phase (the state codes grow with depth)
        // The mesh phases always get their weight in a small scene (its tables fit LDS): the rays walk few objects
        // in much the same order, so the mesh walks of a wave coincide.  In a scene of thousands of instances the
        // rays are scattered over them, and running a handful of mesh lanes first would serialise the wave: there
        // the weight applies only while at least VOTE_MESH_MIN lanes are in a mesh walk.
        const bool mesh_first = Tab::small_scene || c_tri + c_node >= VOTE_MESH_MIN;
        // (the keys are wave-uniform; left to itself the compiler moves them into vector registers for a v_max3 and back)
        uint32_t key = ((c_tri * (mesh_first ? VOTE_W_TRI : 1u)) << 3) | ST_TRI;
        key = smax(key, ((c_node * (mesh_first ? VOTE_W_NODE : 1u)) << 3) | ST_NODE);
        key = smax(key, ((c_tnode * VOTE_W_TNODE) << 3) | ST_TNODE);
        key = smax(key, ((c_obj * VOTE_W_OBJ) << 3) | ST_OBJ);
        key = smax(key, ((c_fetch * VOTE_W_FETCH) << 3) | ST_FETCH);
        const uint32_t phase = key & 7u;

        if (MESHLOC != MESH_NONE && (phase == ST_TRI || phase == ST_NODE)) {
          // The two mesh phases alternate many times between two steps of any other phase, and they change only a few of
          // the lane's registers.  They run in a loop of their own, which is left when the full vote could pick another
          // phase: inside it the rest of the lane state is loop-invariant (no copies at the loop's joins) and the vote
          // counts two states instead of five.  The sequence of phases is the one the full vote would produce.
          uint32_t ph = phase, cn = c_node;
          // (lanes parked without a ray take no part in any vote, and none parks inside this loop)
          const uint32_t c_off = IO::MEGA ? wave_count(st == ST_IDLE || st == ST_DONE) : 0u;
          for (;;) {
            if (ph == ST_TRI) tri_phase(); else node_phase(cn);
            const uint32_t ct2 = wave_count(st == ST_TRI);
            cn = wave_count(st == ST_NODE);
            const bool mf = Tab::small_scene || ct2 + cn >= VOTE_MESH_MIN;
            const uint32_t kt = ct2 * (mf ? VOTE_W_TRI : 1u), kn = cn * (mf ? VOTE_W_NODE : 1u);
            const uint32_t others = (64u - ct2 - cn - c_off) * VOTE_W_OTHER_MAX;   // upper bound of any other phase's key
            const uint32_t km = kt >= kn ? kt : kn;
            if (km == 0u || km < others) break;   // equal keys go to the deeper phase: a mesh phase
            DIAG_END(ph)
            DIAG_TOP
            TL_TRIP
            ph = kt >= kn ? ST_TRI : ST_NODE;
          }
          DIAG_END(ph)
        }
        if (phase == ST_TNODE) {
          // (a tree of meshes: the lanes that leave for an object are gone for a whole mesh walk, the others are worth waiting
          // for down to 1/4: instanced grid +8 %; a tree of spheres: 3/4, the objects are done in one OBJ step)
          constexpr uint32_t tnode_stay_num = MESHLOC == MESH_NONE ? TNODE_STAY_NUM : TNODE_STAY_NUM_MESH;
          // The phase exists twice: with the reference's box test (node_passes) while some lane inside an object tree needs its
          // NaN rules (tree_exact), and with the plain one (node_passes_plain) otherwise: one wave-uniform branch per phase.
          auto tnode_phase = [&](auto exact_c) {
          constexpr bool EXACT = decltype(exact_c)::value;
          for (uint32_t c = c_tnode; c * STAY_DEN >= c_tnode * tnode_stay_num && c > 0; c = wave_count(st == ST_TNODE)) {
            DIAG_STEP(ST_TNODE, c)
            constexpr bool TNODE_PAIRS = TNODE_PAIRS_ON && MESHLOC != MESH_NONE;
            if (ORD_T && st == ST_TNODE && (fl & F_TORD)) {
                // nearer child first (ordered_step): t_n doubles as the end of a subtree walked in the reference's order
                uint32_t spt = (fl >> SPT_SHIFT) & SP_MASK;
                bool range = (fl & F_TRANGE) != 0u, leaf, done;
                uint32_t first = 0u, end = 0u;
                ordered_step<EXACT, TC>(S.nodes + 2 * (size_t)t_base, S.wide + 4 * (size_t)t_base, wpos, winv, wtmax, t_i, t_n, spt, cap_t, range, ostack + (TC ? lane : 2u * lane),
                                    leaf, first, end, done);
                fl = (fl & ~(F_TRANGE | (SP_MASK << SPT_SHIFT))) | (range ? F_TRANGE : 0u) | (spt << SPT_SHIFT);
                o_cur = leaf ? t_first + first : o_cur; o_end = leaf ? t_first + end : o_end;
                fl &= done ? ~(F_INTREE | F_TORD) : ~0u;
                st = (leaf || done) ? ST_OBJ : ST_TNODE;
            }
            if (TNODE_PAIRS && st == ST_TNODE && !(ORD_T && (fl & F_TORD))) {
                // two nodes per step, as node_step2 does for a mesh tree in global memory: the object tree's nodes are read
                // through L2, and the node behind the current one comes with the same round trip
                const uint32_t i2 = t_i + 1u < t_n ? t_i + 1u : t_i;
                const float4* np = S.nodes + 2 * ((size_t)t_base + t_i);
                const float4* mp = S.nodes + 2 * ((size_t)t_base + i2);
                const float4 n0 = np[0], n1 = np[1], m0 = mp[0], m1 = mp[1];
                const bool sx = signbit32(winv.x), sy = signbit32(winv.y), sz = signbit32(winv.z);
                const float tm = walk_tmax();
                const uint32_t A1 = __float_as_uint(n1.z), B1 = __float_as_uint(n1.w), A2 = __float_as_uint(m1.z), B2 = __float_as_uint(m1.w);
                const bool leaf1 = (A1 & LEAF_BIT) != 0u, leaf2 = (A2 & LEAF_BIT) != 0u;
                const uint32_t first1 = A1 & ~LEAF_BIT, first2 = A2 & ~LEAF_BIT;
                const bool pass1 = EXACT ? node_passes(n0, n1, wpos, winv, sx, sy, sz, tm) : node_passes_plain(n0, n1, wpos, winv, tm);
                const bool pass2 = EXACT ? node_passes(m0, m1, wpos, winv, sx, sy, sz, tm) : node_passes_plain(m0, m1, wpos, winv, tm);
                const bool enter1 = pass1 && leaf1 && first1 < B1;
                const bool on = !enter1 && (pass1 || leaf1) && t_i + 1u < t_n;
                const bool enter2 = on && pass2 && leaf2 && first2 < B2;
                const uint32_t next1 = (pass1 || leaf1) ? t_i + 1u : A1;
                const uint32_t next2 = enter2 ? t_i + 1u : ((pass2 || leaf2) ? t_i + 2u : A2);
                const bool enter = enter1 || enter2;
                o_cur = enter1 ? t_first + first1 : (enter2 ? t_first + first2 : o_cur);
                o_end = enter1 ? t_first + B1 : (enter2 ? t_first + B2 : o_end);
                t_i = enter1 ? t_i : (on ? next2 : next1);
                bool tree_done = !enter && t_i >= t_n;
                if (tree_done && walk_again()) tree_done = false;
                fl &= tree_done ? ~F_INTREE : ~0u;
                st = (enter || tree_done) ? ST_OBJ : ST_TNODE;
            }
            if (!TNODE_PAIRS && st == ST_TNODE && !(ORD_T && (fl & F_TORD))) {
                const float4* np = S.nodes + 2 * ((size_t)t_base + t_i);
                float4 n0 = np[0], n1 = np[1];
                const uint32_t A = __float_as_uint(n1.z), B = __float_as_uint(n1.w);
                const bool leaf = (A & LEAF_BIT) != 0u;
                const uint32_t first = A & ~LEAF_BIT;
                const bool pass = EXACT ? node_passes(n0, n1, wpos, winv, signbit32(winv.x), signbit32(winv.y), signbit32(winv.z), walk_tmax())
                                        : node_passes_plain(n0, n1, wpos, winv, walk_tmax());
                const bool enter = pass && leaf && first < B;
                o_cur = enter ? t_first + first : o_cur; o_end = enter ? t_first + B : o_end;
                t_i = enter ? t_i : ((pass || leaf) ? t_i + 1 : A);
                bool tree_done = !enter && t_i >= t_n;
                if (tree_done && walk_again()) tree_done = false;
                fl &= tree_done ? ~F_INTREE : ~0u;
                st = (enter || tree_done) ? ST_OBJ : ST_TNODE;
            }
          }
          };
          if (!PLAIN_T || tree_exact) tnode_phase(std::true_type{}); else tnode_phase(std::false_type{});
        }
        if (phase == ST_OBJ) {
          // A lane stays in this phase across objects that end at once: a primitive that is not a mesh, or a mesh
          // whose root box the ray misses (most instances, for most rays): its root is tested right here.
          // (scenes without meshes: the OBJ step is the leaf work of the object tree and leaves early, at 1/2 of its lanes;
          // with meshes most steps end at a missed root box and the phase is worth staying in until no lane is left: bunny
          // scene +1 %, instanced grid +6 %, against -3 % for the sphere scene)
          constexpr uint32_t obj_stay_num = MESHLOC == MESH_NONE ? OBJ_STAY_NUM : OBJ_STAY_NUM_MESH;
          for (uint32_t c = c_obj; c * STAY_DEN >= c_obj * obj_stay_num && c > 0; c = wave_count(st == ST_OBJ)) {
            DIAG_STEP(ST_OBJ, c)
            DIAG_SUB_BEGIN
            if (st == ST_OBJ) {
                // (1) fold a finished mesh walk into the ray's closest hit (bvh.rs:262-264, transform.rs:107)
                if (MESHLOC != MESH_NONE && (fl & (F_PEND | F_FOUND)) == (F_PEND | F_FOUND)) {
                    float t_out = otmax * oscale;
                    if (GENERIC_XF && (fl & F_GENXF)) { const ObjHead ob = tab.object(oi); t_out = ops_t_out(tab, ob.op_first, ob.op_count, otmax); }
                    // (an ordered tree walk can reach an object after one that follows it in tree order: at the same t the
                    // reference keeps the later one, aggregate.rs:28-31)
                    if (!(ORD_T && (fl & F_TORD) && best.obj != MISS && t_out == wtmax && oi < best.obj)) {
                        best.obj = oi; best.prim = oprim; best.t_local = otmax; best.u = ou; best.v = ov;
                        wtmax = t_out;
                    }
                }
                fl &= ~(F_PEND | F_FOUND);
                // (2) where does the next object come from?
                if (o_cur == o_end) {
                    if (ORD_T && (fl & F_TORD)) {   // leaf of an ordered walk finished
                        bool range = (fl & F_TRANGE) != 0u;
                        uint32_t spt = (fl >> SPT_SHIFT) & SP_MASK;
                        const bool tree_done = ordered_leaf_done<TC>(t_i, t_n, spt, range, ostack + (TC ? lane : 2u * lane), wtmax);
                        fl = (fl & ~(F_TRANGE | (SP_MASK << SPT_SHIFT))) | (range ? F_TRANGE : 0u) | (spt << SPT_SHIFT);
                        if (tree_done) fl &= ~(F_INTREE | F_TORD); else st = ST_TNODE;
                    } else if (fl & F_INTREE) {  // leaf of the object tree finished: i += 1
                        t_i++;
                        if (t_i >= t_n && !walk_again()) fl &= ~F_INTREE; else st = ST_TNODE;
                    }
                    if (!(fl & F_INTREE) && st == ST_OBJ) {
                        if (root_i >= S.n_root) {
                            fl |= F_HAVE; st = ST_FETCH;  // GeometryList exhausted: the ray is finished
                        } else {
                            const DevRootItem it = tab.root(root_i++);
                            if (it.kind == ROOT_OBJECT) { o_cur = it.index; o_end = it.index + 1; }
                            else {
                                const DevTree tr = tab.tree(it.index);
                                t_base = tr.node_base; t_n = tr.n_nodes; t_first = tr.first_object; t_i = 0;
                                if (!S.any_box) winv = mk(1.0f / wdir.x, 1.0f / wdir.y, 1.0f / wdir.z);  // bvh.rs:220
                                if (PLAIN_T) fl = plain_ray(wpos, winv) ? (fl & ~F_ODDW) : (fl | F_ODDW);
                                if (t_n > 0) {
                                    fl |= F_INTREE; st = ST_TNODE;
                                    if (ORD_T && ord_t && (tr.deepen & TREE_ORDERED)) {
                                        fl = (fl & ~(F_TRANGE | (SP_MASK << SPT_SHIFT))) | F_TORD;
                                        t_i = NI_ROOT; t_lim = __builtin_inff();
                                    } else walk_begin(tr.deepen);
                                }
                            }
                        }
                    }
                }
                DIAG_SUB_END(3)
                DIAG_SUB_MARK
                // (3) enter the next object
                if (st == ST_OBJ && o_cur < o_end) {
                    // the object's box in the space of the ray: most rays miss most objects, and this test needs neither
                    // the wrappers nor a reciprocal direction per object.  It only ever skips what the object's own first
                    // test would turn away (the comparisons are false for NaN, so nothing is skipped on one).
                    // Large scenes read the object's 32-byte instance record and take the rest from its class in LDS.
                    // A run of objects that are skipped this way is passed in one step: a step is a trip of the whole wave
                    // through vote and phase, the test of one more box a dozen instructions (headline frame +0.8 %).
                    const float tm_in = (fl & F_INTREE) ? walk_tmax() : wtmax;   // t_max the object is entered with
                    ObjHead ob;
                    InstRec in;
                    bool classed, skip;
                    for (;;) {
                        oi = o_cur++;
                        in = tab.inst(oi);
                        classed = Tab::has_inst && in.cls != CLS_NONE;
                        if (classed) { ob = tab.cls_head(in.cls); ob.scale = in.scale; ob.inv_scale = in.inv_scale; }
                        else ob = tab.object(oi);
                        skip = false;
                        if (ob.xf_kind & XF_BOX) {
                            ObjBox ox;
                            if (classed) { ox = tab.cls_box(in.cls); ox.mn = ox.mn + in.off; ox.mx = ox.mx + in.off; }
                            else ox = tab.object_box(oi);
                            // The slab roots carry a rounding error of a few ulps of the largest plane distance, which for a far
                            // origin exceeds the fixed pad of the box: the interval is widened by that much before it is judged.
                            const f3 pa = (ox.mn - wpos) * winv, pb = (ox.mx - wpos) * winv;
                            const float bt0 = max2(min2(pa.x, pb.x), max2(min2(pa.y, pb.y), min2(pa.z, pb.z)));
                            const float bt1 = min2(max2(pa.x, pb.x), min2(max2(pa.y, pb.y), max2(pa.z, pb.z)));
                            const float err = 1e-6f * fmaxf(fabsf(bt0), fabsf(bt1)) + ox.slack;   // NaN or inf roots: nothing is skipped
                            skip = (bt0 > bt1 + err) || (bt0 > tm_in * 1.00001f + err) || (bt1 < -err);
                        }
                        // (scenes without meshes flag no object for the box test: there the loop only costs, 1.5 % of the sphere frame)
                        if (!(RTAMD_SKIP_RUN != 0 && MESHLOC != MESH_NONE && skip && o_cur < o_end)) break;
                    }
                    if (!skip) {
                    ObjXf xf;   // read with the head: one round trip instead of two
                    if (classed) { xf = tab.cls_xf(in.cls); xf.off = in.off; }
                    else xf = tab.object_xf(oi);
                    opos = wpos; otmax = tm_in;
                    odir = wdir;
                    oscale = ob.scale;
                    if (GENERIC_XF && (ob.xf_kind & XF_GENERIC)) {
                        fl |= F_GENXF;
                        // a BoundingBox wrapper that turns the ray away: go on as for a mesh without nodes
                        if (!ops_in(tab, ob.op_first, ob.op_count, opos, odir, otmax)) { ob.geom_kind = GEOM_MESH; ob.n_nodes = 0u; }
                    } else {
                        if (GENERIC_XF) fl &= ~F_GENXF;
                        xf_in_pos(ob.xf_kind, xf, ob.inv_scale, opos, otmax);
                        odir = xf_in_dir(ob.xf_kind, xf, wdir);
                    }
                    DIAG_SUB_END(4)
                    DIAG_SUB_MARK
                    if (MESHLOC != MESH_NONE && ob.geom_kind == GEOM_MESH) {
                        n_n = ob.n_nodes; n_i = 0; node_off = ob.node_base; tri_off = ob.tri_base;
                        if (MESHLOC == MESH_MIXED) fl = (ob.geom_index == S.lds_mesh) ? (fl | F_MLDS) : (fl & ~F_MLDS);
                        oinv = mk(1.0f / odir.x, 1.0f / odir.y, 1.0f / odir.z);  // bvh.rs:220
                        // a ray (or a tree) for which a slab product can be a NaN walks its mesh alone, after this phase
                        if (PLAIN_NODE && !(S.plain_boxes != 0u && plain_ray(opos, oinv))) st = ST_WALK;
                        else if (n_n > 0) {
                            node_step();   // the root's own box
                            if (ORD_M && ord_m) {   // behind a root that is an inner node the walk goes on from its two-child record
                                // (occlusion queries too: their t_max is constant, but what occludes a shadow ray tends to lie near
                                // its origin -- leaving them in the reference's order costs 4-15 % of a frame)
                                fl &= ~(F_MRANGE | (SP_MASK << SPM_SHIFT));
                                n_i = st == ST_NODE ? 0u : n_i;
                            }
                        }
                        DIAG_SUB_END(5)
                    } else {
                        float t = 0.0f;
                        bool hit;
                        if (fl & F_ANY) {
                            hit = primitive_any<FEAT>(S, tab, ob, opos, odir, otmax);
                            if (hit) { fl |= F_OCC | F_HAVE; st = ST_FETCH; }
                        } else {
                            float u = 0.f, v = 0.f;
                            hit = primitive_closest<FEAT>(S, tab, ob, opos, odir, otmax, t, u, v);
                            const float t_out = !hit ? 0.0f : ((GENERIC_XF && (fl & F_GENXF)) ? ops_t_out(tab, ob.op_first, ob.op_count, t) : t * ob.scale);
                            if (hit && !(ORD_T && (fl & F_TORD) && best.obj != MISS && t_out == wtmax && oi < best.obj)) {
                                best.obj = oi; best.prim = 0; best.t_local = t; best.u = u; best.v = v;
                                wtmax = t_out;
                            }
                        }
                    }
                    }
                }
            }
          }
        }
        if (PLAIN_NODE && MESHLOC != MESH_NONE && phase == ST_OBJ && wave_count(st == ST_WALK) != 0u) {
            // The boxes of a mesh tree are tested with plain minima and maxima (node_passes_plain), which presumes that no slab
            // product is a NaN.  A ray for which one can be (a direction component of exactly zero, a non-finite origin, a tree
            // with a box that is not finite) needs the reference's NaN rules at every box: such rays are rare, the OBJ phase
            // parks them in ST_WALK, and the lane walks the whole mesh here, by itself, through global memory, as the plain
            // kernels do, before the next vote.  A closest-hit walk also answers an occlusion query: until its first accepted
            // triangle it is the same walk.
            if (st == ST_WALK) {
                const GlobalMem gm{S.nodes + 2 * (size_t)node_off, S.tris + 3 * (size_t)tri_off};
                float tm = otmax, hu = 0.0f, hv = 0.0f;
                uint32_t prim = 0u;
                st = ST_OBJ;
                if (traverse_mesh<false>(gm, n_n, opos, odir, tm, prim, hu, hv)) {
                    if (fl & F_ANY) { fl |= F_OCC | F_HAVE; st = ST_FETCH; }
                    else { otmax = tm; oprim = prim; ou = hu; ov = hv; fl |= F_PEND | F_FOUND; }
                }
            }
        }
        if (PLAIN_T && phase == ST_OBJ && S.plain_tree_boxes != 0u)   // (lanes enter object trees only in the OBJ phase)
            tree_exact = wave_count((fl & (F_INTREE | F_ODDW)) == (F_INTREE | F_ODDW)) != 0u;
        if (IO::MEGA && __builtin_expect(phase == ST_FETCH, 0)) {
            if constexpr (IO::MEGA) {
            DIAG_STEP(ST_FETCH, c_fetch)
            pool_fetch(io, lane, st, fl, ridx, rtag, best, wtmax, pool_rot, n_closest_taken, n_any_taken, [&](const uint32_t (&raw)[8]) {
                bool any;
                io.decode(ridx, raw, wpos, wdir, wtmax, any, rtag);
                if (S.any_box) winv = mk(1.0f / wdir.x, 1.0f / wdir.y, 1.0f / wdir.z);
                best.obj = MISS; best.prim = 0; best.t_local = 0.f; best.u = 0.f; best.v = 0.f;
                const bool run = S.root_first != MISS;
                root_i = run ? S.n_root : 0u; o_cur = run ? S.root_first : 0u; o_end = run ? S.root_first + S.n_root : 0u;
                fl = any ? F_ANY : 0u;
                st = ST_OBJ;
            }, d_pool);
            }
        }
        if (!IO::MEGA && phase == ST_FETCH) {
            DIAG_STEP(ST_FETCH, c_fetch)
            DIAG_SUB_BEGIN
            const bool want = st == ST_FETCH;
            // hand out staged rays to the lanes that want one
            const unsigned long long need = __ballot(want);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));  // wanting lanes below this one
            const uint32_t avail = sg_cnt - sg_pos;
            const uint32_t take = c_fetch < avail ? c_fetch : avail;
            const bool got = want && rank < take;
            uint32_t raw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (take > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the batch was requested in an earlier FETCH step
                const uint32_t src = (sg_pos + rank) & 63u;
#pragma unroll
                for (int c = 0; c < 8; c++) raw[c] = stage[c * 64 + src];
                if (RTAMD_LDS_DUP & 4) {
#pragma unroll
                    for (int c = 0; c < 8; c++) { const uint32_t z = ((volatile lds_u32*)stage)[c * 64 + src]; asm volatile("" ::"v"(z)); }
                }
            }
            DIAG_SUB_END(0)
            { DIAG_SUB_BEGIN
            if (want && (fl & F_HAVE)) {
                if (fl & F_ANY) io.store_any(ridx, rtag, (fl & F_OCC) != 0u);
                else io.store_closest(ridx, rtag, best.obj != MISS, best, wtmax);
            }
            DIAG_SUB_END(2) }
            fl &= want ? ~F_HAVE : ~0u;
            { DIAG_SUB_BEGIN
            if (got) {
                bool any;
                ridx = sg_base + sg_pos + rank;
                io.decode(ridx, raw, wpos, wdir, wtmax, any, rtag);
                if (S.any_box) winv = mk(1.0f / wdir.x, 1.0f / wdir.y, 1.0f / wdir.z);
                best.obj = MISS; best.prim = 0; best.t_local = 0.f; best.u = 0.f; best.v = 0.f;
                // a root list that is one run of objects is walked as a range (no list reads)
                const bool run = S.root_first != MISS;
                root_i = run ? S.n_root : 0u; o_cur = run ? S.root_first : 0u; o_end = run ? S.root_first + S.n_root : 0u;
                fl = any ? F_ANY : 0u;
                st = ST_OBJ;
            }
            DIAG_SUB_END(6) }
            sg_pos += take;
            // staged batch used up: request the next one now, it is read in a later FETCH step
            { DIAG_SUB_BEGIN
            if (sg_pos == sg_cnt && !exhausted) {
                if (chunk_next == chunk_end && ks < k_rounds) {  // next batch of the static deal
                    const bool tail = ks >= k_static;          // (the shorter batches behind the full ones)
                    const uint32_t len = tail ? tail_bs : bs;
                    const uint64_t b = tail ? (uint64_t)k_static * n_waves * bs + (uint64_t)wave_id * tail_bs : ((uint64_t)wave_id + (uint64_t)ks * n_waves) * bs;
                    ks++;
                    if (b < n_items) { chunk_next = (uint32_t)b; chunk_end = (n_items - chunk_next) < len ? n_items : chunk_next + len; }
                    else ks = k_rounds;
                }
                if (chunk_next == chunk_end) {                   // shared remainder
                    uint32_t base = n_items;
                    if (dyn_base < n_items) {
                        DIAG_SUB_BEGIN
                        const uint32_t got = claim_chunk(work_head, chunk);
                        DIAG_SUB_END(1)
                        base = got < n_items - dyn_base ? dyn_base + got : n_items;
                    }
                    if (base >= n_items) {
                        exhausted = true;
                        TL_DRY
                    } else {
                        chunk_next = base;
                        chunk_end = (n_items - base) < chunk ? n_items : base + chunk;
                    }
                }
                sg_pos = 0;
                sg_cnt = 0;
                if (!exhausted) {
                    const uint32_t left = chunk_end - chunk_next;
                    sg_base = chunk_next;
                    sg_cnt = left < bs ? left : bs;
                    chunk_next += sg_cnt;
                    const void* src[8];
                    io.stage_addr(sg_base + (lane < sg_cnt ? lane : sg_cnt - 1u), src);
                    DIAG_SUB_END(7)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads of the old batch have returned
#pragma unroll
                    for (int c = 0; c < 8; c++) stage_row(src[c], stage_lds + (uint32_t)c * 256u);
                }
            }
            }
            if (st == ST_FETCH && exhausted && sg_cnt == 0) st = ST_DONE;
        }
        DIAG_END(phase)
    }
    if constexpr (IO::MEGA) {
        // (every pixel must have come to its end)
        uint32_t left = 0;
#pragma unroll
        for (uint32_t q = 0; q < POOL_PER_LANE; q++) left |= ((volatile lds_u32*)io.pix)[POOL_PER_LANE * lane + q] & PW_ALIVE;
        if (__ballot(left != 0u) != 0ull) io.fail(lane, 2u);
        io.finish(lane, n_closest_taken, n_any_taken);
    }
    DIAG_FLUSH
    TL_FLUSH
}

}  // namespace rtamd
