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
// staged in registers (lane l holds ray batch_base + l; a lane that needs a ray pulls it across
// lanes with ds_bpermute).  The batch after it is requested the moment the staged one is used up,
// and the queue head is advanced one chunk ahead of need, so neither the ray loads nor the atomic
// are waited for in the step that issues them: their latency overlaps the traversal phases.
#pragma once
#include "rt_device.h"

namespace rtamd {

enum : uint32_t { ST_FETCH = 0, ST_OBJ = 1, ST_TNODE = 2, ST_NODE = 3, ST_TRI = 4, ST_DONE = 5 };

// LDS copy of mesh S.lds_mesh, addressed through LDS-typed pointers so that the loads are
// ds_read_b128 (a generic pointer that may be LDS or global compiles to slow flat loads).
struct LdsMesh {
    lds_float4* nodes;
    lds_float4* tris;
};
// Where the mesh trees of a launch live: decided on the host per scene.
enum : int { MESH_GLOBAL = 0, MESH_LDS = 1, MESH_MIXED = 2 };

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

static constexpr uint32_t STAY_NUM = 3, STAY_DEN = 4;

// Diagnostic build only (-DRTAMD_DIAG): per-phase cycle, step and active-lane totals of the scheduler,
// accumulated per wave and added to a global table when the wave ends.  Never compiled into librtamd.so.
#ifdef RTAMD_DIAG
__device__ unsigned long long g_diag[32];
#define DIAG_DECL unsigned long long d_cyc[6] = {0, 0, 0, 0, 0, 0}, d_steps[6] = {0, 0, 0, 0, 0, 0}, d_lanes[6] = {0, 0, 0, 0, 0, 0}; unsigned long long d_t0 = 0;
#define DIAG_TOP d_t0 = __builtin_amdgcn_s_memtime();
#define DIAG_STEP(ph, c) { d_steps[ph] += 1; d_lanes[ph] += (c); }
#define DIAG_END(ph) { d_cyc[ph] += __builtin_amdgcn_s_memtime() - d_t0; }
#define DIAG_FLUSH if (lane == 0) { for (int q = 0; q < 5; q++) { atomicAdd(&g_diag[q * 3], d_cyc[q]); atomicAdd(&g_diag[q * 3 + 1], d_steps[q]); atomicAdd(&g_diag[q * 3 + 2], d_lanes[q]); } }
#else
#define DIAG_DECL
#define DIAG_TOP
#define DIAG_STEP(ph, c)
#define DIAG_END(ph)
#define DIAG_FLUSH
#endif

// Claims the next chunk of the work queue; the returned value is valid in lane 0 only and should be
// read (wave_first) as late as possible so that the atomic's latency is hidden.
__device__ __forceinline__ uint32_t claim_chunk(uint32_t* work_head, uint32_t chunk) {
    uint32_t base = 0;
    if (lane_id() == 0) base = atomicAdd(work_head, chunk);
    return base;
}

// IO concept:
//   void stage(uint32_t idx, uint32_t (&raw)[8]) const;   // issue the loads of ray idx; no branches around loads
//   void decode(uint32_t idx, const uint32_t (&raw)[8], f3& pos, f3& dir, float& tmax, bool& any, uint32_t& tag) const;
//   void store_closest(uint32_t idx, uint32_t tag, bool found, const Best& b, float t_world) const;
//   void store_any(uint32_t idx, uint32_t tag, bool occluded) const;
// `tag` is an opaque word the IO wants back when the result is stored (loaded with the ray, so that
// storing never needs a dependent read).  stage() must load straight into raw[] in uniform control
// flow: a load behind a divergent branch lands in a temporary that is merged right away, which puts
// the wait for it next to the load.
template <int MESHLOC, class Tab, class IO>
__device__ __forceinline__ void trace_scheduler(const DevScene& S, const Tab& tab, const LdsMesh lds, const IO& io, uint32_t* work_head,
                                                const uint32_t n_items, const uint32_t chunk, uint32_t head_claim) {
    const uint32_t lane = lane_id();
    const unsigned long long lt_mask = (1ULL << lane) - 1ULL;

    uint32_t st = ST_FETCH;
    bool have = false, any = false, occluded = false;
    uint32_t ridx = 0, rtag = 0;
    f3 wpos = mk(0, 0, 0), wdir = mk(0, 0, 1), winv = mk(0, 0, 0);
    bool wngx = false, wngy = false, wngz = false;
    float wtmax = 0.0f;
    // staged batch: lane l holds the raw words of ray sg_base + l, l < sg_cnt; sg_pos rays are handed out
    uint32_t sg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t sg_base = 0, sg_cnt = 0, sg_pos = 0;  // wave-uniform
    Best best;
    best.obj = MISS; best.prim = 0; best.t_local = 0.f; best.u = 0.f; best.v = 0.f;
    uint32_t root_i = 0;
    bool in_tree = false;
    uint32_t t_i = 0, t_n = 0, t_base = 0, t_first = 0;
    uint32_t o_cur = 0, o_end = 0;
    uint32_t oi = 0, op_first = 0, op_count = 0;
    f3 opos = mk(0, 0, 0), odir = mk(0, 0, 1), oinv = mk(0, 0, 0);
    bool ongx = false, ongy = false, ongz = false;
    float otmax = 0.0f;
    uint32_t n_i = 0, n_n = 0, node_off = 0, tri_off = 0;
    bool mesh_lds = false, mesh_pending = false, ofound = false;
    uint32_t k_cur = 0, k_end = 0, oprim = 0;
    float ou = 0.f, ov = 0.f;
    uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform
    bool exhausted = false;                   // wave-uniform: the global queue has run dry

    DIAG_DECL
    for (;;) {
        DIAG_TOP
        // ---- vote: run the phase most lanes are waiting for ----
        const uint32_t c_tri = __popcll(__ballot(st == ST_TRI));
        const uint32_t c_node = __popcll(__ballot(st == ST_NODE));
        const uint32_t c_tnode = __popcll(__ballot(st == ST_TNODE));
        const uint32_t c_obj = __popcll(__ballot(st == ST_OBJ));
        const uint32_t c_fetch = __popcll(__ballot(st == ST_FETCH));
        if (c_tri + c_node + c_tnode + c_obj + c_fetch == 0) break;  // every lane is ST_DONE
        uint32_t phase = ST_TRI, top = c_tri;
        if (c_node > top) { top = c_node; phase = ST_NODE; }
        if (c_tnode > top) { top = c_tnode; phase = ST_TNODE; }
        if (c_obj > top) { top = c_obj; phase = ST_OBJ; }
        if (c_fetch > top) { top = c_fetch; phase = ST_FETCH; }

        if (phase == ST_TRI) {
          // stay in the phase while at least STAY_NUM/STAY_DEN of the lanes that entered it still need it:
          // one ballot per step instead of a full vote
          for (uint32_t c = c_tri; c * STAY_DEN >= c_tri * STAY_NUM && c > 0; c = __popcll(__ballot(st == ST_TRI))) {
            DIAG_STEP(ST_TRI, c)
            if (st == ST_TRI) {
                float4 q0, q1, q2;
                if (MESHLOC == MESH_LDS || (MESHLOC == MESH_MIXED && mesh_lds)) {
                    lds_float4* tp = lds.tris + 3 * k_cur;
                    q0 = lds_load4(tp); q1 = lds_load4(tp + 1); q2 = lds_load4(tp + 2);
                } else {
                    const float4* tp = S.tris + 3 * ((size_t)tri_off + k_cur);
                    q0 = tp[0]; q1 = tp[1]; q2 = tp[2];
                }
                float t, u, v;
                if (tri_test(q0, q1, q2, opos, odir, otmax, t, u, v)) {
                    if (any) { occluded = true; have = true; st = ST_FETCH; }
                    else { otmax = t; oprim = k_cur; ou = u; ov = v; ofound = true; }
                }
                k_cur++;
                if (st == ST_TRI && k_cur >= k_end) {  // leaf done: i += 1 (bvh.rs:245)
                    n_i++;
                    if (n_i >= n_n) { mesh_pending = true; st = ST_OBJ; } else st = ST_NODE;
                }
            }
          }
        } else if (phase == ST_NODE) {
          for (uint32_t c = c_node; c * STAY_DEN >= c_node * STAY_NUM && c > 0; c = __popcll(__ballot(st == ST_NODE))) {
            DIAG_STEP(ST_NODE, c)
            if (st == ST_NODE) {
                float4 n0, n1;
                if (MESHLOC == MESH_LDS || (MESHLOC == MESH_MIXED && mesh_lds)) {
                    lds_float4* np = lds.nodes + 2 * n_i;
                    n0 = lds_load4(np); n1 = lds_load4(np + 1);
                } else {
                    const float4* np = S.nodes + 2 * ((size_t)node_off + n_i);
                    n0 = np[0]; n1 = np[1];
                }
                const uint32_t A = __float_as_uint(n1.z), B = __float_as_uint(n1.w);
                const bool leaf = (A & LEAF_BIT) != 0u;
                if (node_passes(n0, n1, opos, oinv, ongx, ongy, ongz, otmax)) {
                    if (leaf && (A & ~LEAF_BIT) < B) { k_cur = A & ~LEAF_BIT; k_end = B; st = ST_TRI; }
                    else n_i++;
                } else {
                    n_i = leaf ? n_i + 1 : A;
                }
                if (st == ST_NODE && n_i >= n_n) { mesh_pending = true; st = ST_OBJ; }
            }
          }
        } else if (phase == ST_TNODE) {
          for (uint32_t c = c_tnode; c * STAY_DEN >= c_tnode * STAY_NUM && c > 0; c = __popcll(__ballot(st == ST_TNODE))) {
            DIAG_STEP(ST_TNODE, c)
            if (st == ST_TNODE) {
                const float4* np = S.nodes + 2 * ((size_t)t_base + t_i);
                float4 n0 = np[0], n1 = np[1];
                const uint32_t A = __float_as_uint(n1.z), B = __float_as_uint(n1.w);
                const bool leaf = (A & LEAF_BIT) != 0u;
                if (node_passes(n0, n1, wpos, winv, wngx, wngy, wngz, wtmax)) {
                    if (leaf && (A & ~LEAF_BIT) < B) { o_cur = t_first + (A & ~LEAF_BIT); o_end = t_first + B; st = ST_OBJ; }
                    else t_i++;
                } else {
                    t_i = leaf ? t_i + 1 : A;
                }
                if (st == ST_TNODE && t_i >= t_n) { in_tree = false; st = ST_OBJ; }
            }
          }
        } else if (phase == ST_OBJ) {
            DIAG_STEP(ST_OBJ, c_obj)
            if (st == ST_OBJ) {
                // (1) fold a finished mesh walk into the ray's closest hit (bvh.rs:262-264, transform.rs:107)
                if (mesh_pending) {
                    mesh_pending = false;
                    if (ofound) {
                        best.obj = oi; best.prim = oprim; best.t_local = otmax; best.u = ou; best.v = ov;
                        wtmax = ops_t_out(tab, op_first, op_count, otmax);
                    }
                }
                // (2) where does the next object come from?
                if (o_cur == o_end) {
                    if (in_tree) {  // leaf of the object tree finished: i += 1
                        t_i++;
                        if (t_i >= t_n) in_tree = false; else st = ST_TNODE;
                    }
                    if (!in_tree && st == ST_OBJ) {
                        if (root_i >= S.n_root) {
                            have = true; st = ST_FETCH;  // GeometryList exhausted: the ray is finished
                        } else {
                            const DevRootItem it = tab.root(root_i++);
                            if (it.kind == ROOT_OBJECT) { o_cur = it.index; o_end = it.index + 1; }
                            else {
                                const DevTree tr = S.trees[it.index];
                                t_base = tr.node_base; t_n = tr.n_nodes; t_first = tr.first_object; t_i = 0;
                                winv = mk(1.0f / wdir.x, 1.0f / wdir.y, 1.0f / wdir.z);  // bvh.rs:220
                                wngx = signbit32(winv.x); wngy = signbit32(winv.y); wngz = signbit32(winv.z);
                                if (t_n > 0) { in_tree = true; st = ST_TNODE; }
                            }
                        }
                    }
                }
                // (3) enter the next object
                if (st == ST_OBJ && o_cur < o_end) {
                    oi = o_cur++;
                    const DevObject ob = tab.object(oi);
                    op_first = ob.op_first; op_count = ob.op_count;
                    opos = wpos; odir = wdir; otmax = wtmax;
                    ops_in(tab, op_first, op_count, opos, odir, otmax);
                    if (ob.geom_kind == GEOM_MESH) {
                        n_n = ob.n_nodes; n_i = 0; node_off = ob.node_base; tri_off = ob.tri_base;
                        mesh_lds = (MESHLOC != MESH_GLOBAL) && (ob.geom_index == S.lds_mesh);
                        oinv = mk(1.0f / odir.x, 1.0f / odir.y, 1.0f / odir.z);  // bvh.rs:220
                        ongx = signbit32(oinv.x); ongy = signbit32(oinv.y); ongz = signbit32(oinv.z);
                        ofound = false;
                        if (n_n > 0) st = ST_NODE;
                    } else {
                        float t = 0.0f;
                        bool hit;
                        if (any) {
                            hit = primitive_any(S, tab, ob, opos, odir, otmax);
                        } else {
                            float u = 0.f, v = 0.f;
                            hit = primitive_closest(S, tab, ob, opos, odir, otmax, t, u, v);
                            if (hit) {
                                best.obj = oi; best.prim = 0; best.t_local = t; best.u = u; best.v = v;
                                wtmax = ops_t_out(tab, op_first, op_count, t);
                            }
                        }
                        if (any && hit) { occluded = true; have = true; st = ST_FETCH; }
                    }
                }
            }
        } else {  // ST_FETCH
            DIAG_STEP(ST_FETCH, c_fetch)
            const bool want = st == ST_FETCH;
            if (want && have) {
                have = false;
                if (any) io.store_any(ridx, rtag, occluded);
                else io.store_closest(ridx, rtag, best.obj != MISS, best, wtmax);
            }
            // hand out staged rays to the lanes that want one
            const unsigned long long need = __ballot(want);
            const uint32_t rank = __popcll(need & lt_mask);
            const uint32_t avail = sg_cnt - sg_pos;
            const uint32_t take = c_fetch < avail ? c_fetch : avail;
            if (take > 0) {
                const uint32_t src = sg_pos + rank;
                uint32_t raw[8];
#pragma unroll
                for (int c = 0; c < 8; c++) raw[c] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)sg[c]);
                if (want && rank < take) {
                    ridx = sg_base + src;
                    io.decode(ridx, raw, wpos, wdir, wtmax, any, rtag);
                    best.obj = MISS; best.prim = 0; best.t_local = 0.f; best.u = 0.f; best.v = 0.f;
                    root_i = 0; in_tree = false; o_cur = 0; o_end = 0;
                    mesh_pending = false; occluded = false; ofound = false;
                    st = ST_OBJ;
                }
                sg_pos += take;
            }
            // staged batch used up: request the next one now, it is consumed in a later FETCH step
            if (sg_pos == sg_cnt && !exhausted) {
                if (chunk_next == chunk_end) {
                    const uint32_t base = __builtin_amdgcn_readfirstlane(head_claim);
                    if (base >= n_items) {
                        exhausted = true;
                    } else {
                        chunk_next = base;
                        chunk_end = (n_items - base) < chunk ? n_items : base + chunk;
                        head_claim = claim_chunk(work_head, chunk);
                    }
                }
                sg_pos = 0;
                sg_cnt = 0;
                if (!exhausted) {
                    const uint32_t left = chunk_end - chunk_next;
                    sg_base = chunk_next;
                    sg_cnt = left < 64u ? left : 64u;
                    chunk_next += sg_cnt;
                    io.stage(sg_base + (lane < sg_cnt ? lane : sg_cnt - 1u), sg);
                }
            }
            if (st == ST_FETCH && exhausted && sg_cnt == 0) st = ST_DONE;
        }
        DIAG_END(phase)
    }
    DIAG_FLUSH
}

}  // namespace rtamd
