// LDS staging of the scene for the scheduler kernels (shared by the kernel files).
#pragma once
#include "rt_sched.h"

namespace rtamd {

extern __shared__ float4 g_lds[];

// Stages the LDS-resident parts of the scene for a scheduler kernel: the small mesh (MESHLOC != GLOBAL) followed by the
// scene tables.  LDS layout, in float4 units:
//   TAB_LDS: [mesh nodes | mesh triangles | objects (8 each) | ops (3 each) | cuboids (3 each) | root items (uint2)]
//   TAB_MID: [mesh nodes | mesh triangles | classes (8 each) | cuboids (3 each) | tree headers (uint4) | root items (uint2)]
enum : int { TAB_GLOBAL = 0, TAB_LDS = 1, TAB_MID = 2 };
// Block-wide copy global -> LDS in two halves, so that the loads of several tables are all in flight before the first
// store waits for one: `begin` issues up to D loads per thread, `end` stores them (and copies whatever lies beyond D
// elements per thread one by one).  A trace launch starts with this copy in every block; as one load and one store per
// loop trip it was a chain of a dozen memory round trips.
template <int D, class T>
__device__ __forceinline__ void block_copy_begin(T (&v)[D], const T* src, uint32_t n) {
    if (n == 0u) {   // (nothing to read: src may be null)
#pragma unroll
        for (int k = 0; k < D; k++) v[k] = T{};
        return;
    }
#pragma unroll
    for (int k = 0; k < D; k++) {
        // (threads past the end re-read element 0: an unconditional load keeps v in registers)
        const uint32_t i = threadIdx.x + (uint32_t)k * blockDim.x;
        v[k] = src[i < n ? i : 0u];
    }
}
template <int D, class T>
__device__ __forceinline__ void block_copy_end(const T (&v)[D], T* dst, const T* src, uint32_t n) {
#pragma unroll
    for (int k = 0; k < D; k++) {
        const uint32_t i = threadIdx.x + (uint32_t)k * blockDim.x;
        if (i < n) dst[i] = v[k];
    }
    for (uint32_t i = threadIdx.x + (uint32_t)D * blockDim.x; i < n; i += blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ uint32_t n_node4_of(const DevScene& S) { return 2u * S.meshes[S.lds_mesh].n_nodes; }
template <int MESHLOC, int TABKIND>
__device__ __forceinline__ void stage_scene(const DevScene& S, LdsMesh& lds, LdsTab& ltab, MidTab& mtab) {
    uint32_t base = 0;  // float4 offset
    lds.nodes = (lds_float4*)g_lds;
    lds.tris = (lds_float4*)g_lds;
    lds.tris1 = (lds_float4*)g_lds;
    lds.trisz = (__attribute__((address_space(3))) const float*)g_lds;
    constexpr bool MESH = MESHLOC == MESH_LDS || MESHLOC == MESH_MIXED;
    uint32_t n_tris = 0;
    // the staged mesh's triangles go to three planes (see LdsMesh): quad c of triangle k lands in plane c
    auto store_tris = [&](const float4 (&v)[8], const float4* src, uint32_t n4) {
        float4* q0 = g_lds + n_node4_of(S);
        float4* q1 = q0 + n_tris;
        float* z = (float*)(q1 + n_tris);
        auto put = [&](uint32_t i, const float4& q) {
            const uint32_t k = i / 3u, c = i - 3u * k;
            if (c == 0u) q0[k] = q; else if (c == 1u) q1[k] = q; else z[k] = q.x;
        };
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = threadIdx.x + (uint32_t)k * blockDim.x;
            if (i < n4) put(i, v[k]);
        }
        for (uint32_t i = threadIdx.x + 8u * blockDim.x; i < n4; i += blockDim.x) put(i, src[i]);
    };
    float4 vn[2], vt[8], va[1], vb[1], vc[1];
    uint32_t n_node4 = 0, n_tri4 = 0;
    const float4 *gn = nullptr, *gt = nullptr;
    if (MESH) {
        const DevMesh m = S.meshes[S.lds_mesh];
        n_node4 = 2 * m.n_nodes;
        n_tri4 = S.lds_bytes / 16 - n_node4;   // (lds_bytes: the mesh as 32-byte nodes + 48-byte triangles, as global memory holds it)
        n_tris = n_tri4 / 3u;
        gn = S.nodes + 2 * (size_t)m.node_base;
        gt = S.tris + 3 * (size_t)m.tri_base;
        block_copy_begin(vn, gn, n_node4);
        block_copy_begin(vt, gt, n_tri4);
        lds.tris = (lds_float4*)g_lds + n_node4;
        lds.tris1 = lds.tris + n_tris;
        lds.trisz = (__attribute__((address_space(3))) const float*)(lds.tris1 + n_tris);
        base = lds_mesh_bytes(m.n_nodes, n_tris) / 16u;
    }
    if (TABKIND == TAB_LDS) {
        const uint32_t n_obj4 = 8 * S.n_objects, n_op4 = 3 * S.n_ops, n_cub4 = 3 * S.n_cuboids;
        const float4* go = (const float4*)S.objects;
        const float4* gp = (const float4*)S.ops;
        const float4* gc = (const float4*)S.cuboids;
        block_copy_begin(va, go, n_obj4);
        block_copy_begin(vb, gp, n_op4);
        block_copy_begin(vc, gc, n_cub4);
        uint2* lr = (uint2*)(g_lds + base + n_obj4 + n_op4 + n_cub4);
        for (uint32_t i = threadIdx.x; i < S.n_root; i += blockDim.x) lr[i] = make_uint2(S.root[i].kind, S.root[i].index);
        if (MESH) { block_copy_end(vn, g_lds, gn, n_node4); store_tris(vt, gt, n_tri4); }
        block_copy_end(va, g_lds + base, go, n_obj4);
        block_copy_end(vb, g_lds + base + n_obj4, gp, n_op4);
        block_copy_end(vc, g_lds + base + n_obj4 + n_op4, gc, n_cub4);
        ltab.S = &S;
        ltab.objects = (lds_float4*)g_lds + base;
        ltab.ops = (lds_float4*)g_lds + base + n_obj4;
        ltab.cuboids = (lds_float4*)g_lds + base + n_obj4 + n_op4;
        ltab.roots = (__attribute__((address_space(3))) const uint32_t*)(g_lds + base + n_obj4 + n_op4 + n_cub4);
    } else if (TABKIND == TAB_MID) {
        const uint32_t n_cls4 = 8 * S.n_classes, n_cub4 = 3 * S.n_cuboids, n_tree4 = S.n_trees;
        const float4* gk = (const float4*)S.classes;
        const float4* gc = (const float4*)S.cuboids;
        const float4* gh = (const float4*)S.trees;
        block_copy_begin(va, gk, n_cls4);
        block_copy_begin(vb, gc, n_cub4);
        block_copy_begin(vc, gh, n_tree4);
        uint2* lr = (uint2*)(g_lds + base + n_cls4 + n_cub4 + n_tree4);
        for (uint32_t i = threadIdx.x; i < S.n_root; i += blockDim.x) lr[i] = make_uint2(S.root[i].kind, S.root[i].index);
        if (MESH) { block_copy_end(vn, g_lds, gn, n_node4); store_tris(vt, gt, n_tri4); }
        block_copy_end(va, g_lds + base, gk, n_cls4);
        block_copy_end(vb, g_lds + base + n_cls4, gc, n_cub4);
        block_copy_end(vc, g_lds + base + n_cls4 + n_cub4, gh, n_tree4);
        mtab.S = &S;
        mtab.classes = (lds_float4*)g_lds + base;
        mtab.cuboids = (lds_float4*)g_lds + base + n_cls4;
        mtab.trees = (__attribute__((address_space(3))) const uint32_t*)(g_lds + base + n_cls4 + n_cub4);
        mtab.roots = (__attribute__((address_space(3))) const uint32_t*)(g_lds + base + n_cls4 + n_cub4 + n_tree4);
    } else if (MESH) {
        block_copy_end(vn, g_lds, gn, n_node4);
        store_tris(vt, gt, n_tri4);
    }
    if (MESH || TABKIND != TAB_GLOBAL) __syncthreads();
}

// Which template instance a launch needs: all mesh trees staged in LDS, none, or a mix.
static inline int mesh_location(const DevScene& S, const LaunchCfg& cfg) {
    if (S.n_meshes == 0) return MESH_NONE;
    if (!cfg.use_lds || S.lds_mesh == MISS) return MESH_GLOBAL;
    return S.n_meshes == 1 ? MESH_LDS : MESH_MIXED;
}
#define RTAMD_DISPATCH_LEAN(KERNEL, ML, TB, lean, grid, lds_bytes, st, ...)                                                    \
    do {                                                                                                                    \
        if (lean) hipLaunchKernelGGL((KERNEL<ML, TB, true>), dim3(grid), dim3(TRACE_BLOCK), lds_bytes, st, __VA_ARGS__);      \
        else hipLaunchKernelGGL((KERNEL<ML, TB, false>), dim3(grid), dim3(TRACE_BLOCK), lds_bytes, st, __VA_ARGS__);          \
    } while (0)
#define RTAMD_DISPATCH_TAB(KERNEL, ML, tab, lean, grid, lds_bytes, st, ...)                                                    \
    do {                                                                                                                    \
        if (tab == TAB_LDS) RTAMD_DISPATCH_LEAN(KERNEL, ML, TAB_LDS, lean, grid, lds_bytes, st, __VA_ARGS__);                 \
        else if (tab == TAB_MID) RTAMD_DISPATCH_LEAN(KERNEL, ML, TAB_MID, lean, grid, lds_bytes, st, __VA_ARGS__);            \
        else RTAMD_DISPATCH_LEAN(KERNEL, ML, TAB_GLOBAL, lean, grid, lds_bytes, st, __VA_ARGS__);                             \
    } while (0)
// (lean_: the scene holds nothing beyond what the lean form of its kernel instance is compiled for, see sched_features)
#define RTAMD_DISPATCH_MESHLOC(S, cfg, KERNEL, grid, st, ...)                                                                 \
    do {                                                                                                                    \
        const int loc_ = mesh_location(S, cfg);                                                                             \
        const int tab_ = !cfg.use_lds ? TAB_GLOBAL : ((S).tab_bytes > 0 ? TAB_LDS : ((S).mid_bytes > 0 ? TAB_MID : TAB_GLOBAL));  \
        const uint32_t lb_ = ((loc_ == MESH_LDS || loc_ == MESH_MIXED) ? (S).lds_sched_bytes : 0u) + (tab_ == TAB_LDS ? (S).tab_bytes : (tab_ == TAB_MID ? (S).mid_bytes : 0u)); \
        const bool lean_ = cfg.lean && ((S).features & ~sched_features(loc_, true)) == 0u;                                    \
        if (loc_ == MESH_NONE) RTAMD_DISPATCH_TAB(KERNEL, MESH_NONE, tab_, lean_, grid, lb_, st, __VA_ARGS__);                \
        else if (loc_ == MESH_LDS) RTAMD_DISPATCH_TAB(KERNEL, MESH_LDS, tab_, lean_, grid, lb_, st, __VA_ARGS__);             \
        else if (loc_ == MESH_MIXED) RTAMD_DISPATCH_TAB(KERNEL, MESH_MIXED, tab_, lean_, grid, lb_, st, __VA_ARGS__);         \
        else RTAMD_DISPATCH_TAB(KERNEL, MESH_GLOBAL, tab_, lean_, grid, lb_, st, __VA_ARGS__);                                \
    } while (0)

}  // namespace rtamd
