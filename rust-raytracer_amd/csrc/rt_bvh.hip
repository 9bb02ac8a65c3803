// GPU build of the reference's binned-SAH tree (src/geom/bvh.rs:10-171), bit-identical to the sequential builder:
// same node boxes, same skip links, same order of the primitives inside every leaf.
//
// The reference builds depth first; the result does not depend on that order, only on what happens inside one node:
// bounds of the range, 16-bucket binning of the centroids per axis, the first strictly cheapest of the 45 candidate
// planes, an in-place Hoare partition ("centroid >= plane" goes left) and the split test.  Here the tree grows level by
// level: one workgroup owns one node (range of the primitive order), all nodes of a level are in flight together, and
// the pre-order numbering with its skip links is derived afterwards from the subtree sizes.
//
// What makes the result identical and not merely equivalent:
//   * every float expression is written as the reference writes it and compiled without contraction;
//   * min / max folds are order independent for finite inputs (the fold of `a < b ? a : b`), so a parallel reduction gives
//     the sequential value (inputs must be finite; a range whose minima mix +0 and -0 may differ in the sign of a zero);
//   * the Hoare loop of bvh.rs:74-94 pairs the k-th misplaced element from the left with the k-th misplaced element from
//     the right: the pairing is computed with prefix counts and applied as independent swaps;
//   * the partition is kept when the node stays a leaf, as in the reference (it partitions before it decides).
#include <hip/hip_runtime.h>

#include <vector>

#include "rt_kernels.h"
#include "rtamd.h"

namespace rtamd {
namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t SMALL_SEGMENT = 512;   // ranges up to this size are built by 64-thread workgroups

struct BSeg { uint32_t begin, end, node, pad; };
struct BNode { float mn[3], mx[3]; uint32_t begin, left, right, size, pre, pad; };   // left / right: node indices in creation order

__device__ __forceinline__ float sel_min(float a, float b) { return a < b ? a : b; }   // min!, common.rs:43-53
__device__ __forceinline__ float sel_max(float a, float b) { return a > b ? a : b; }   // max!, common.rs:32-41
// float <-> unsigned with the same order (for integer atomics in LDS)
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }
constexpr uint32_t ORD_POS_INF = 0xFF800000u;   // f2ord(+inf): identity of the minimum (an empty box, aabb.rs:12-17)
constexpr uint32_t ORD_NEG_INF = 0x007FFFFFu;   // f2ord(-inf): identity of the maximum

__device__ __forceinline__ uint32_t bucket_of(float ratio) {   // `(ratio * 16.0).floor() as usize` then .min(15): saturating, NaN -> 0
    const float f = floorf(ratio * 16.0f);
    if (!(f == f) || f <= 0.0f) return 0u;
    if (f >= 15.0f) return 15u;
    return (uint32_t)f;
}

struct Box6 { float mn[3], mx[3]; };
__device__ __forceinline__ Box6 empty_box() {
    Box6 b;
    for (int k = 0; k < 3; k++) { b.mn[k] = __builtin_huge_valf(); b.mx[k] = -__builtin_huge_valf(); }
    return b;
}
__device__ __forceinline__ Box6 load_box(const float* boxes, uint32_t prim) {
    const float* p = boxes + 6 * (size_t)prim;
    Box6 b;
    b.mn[0] = p[0]; b.mn[1] = p[1]; b.mn[2] = p[2]; b.mx[0] = p[3]; b.mx[1] = p[4]; b.mx[2] = p[5];
    return b;
}
__device__ __forceinline__ void grow(Box6& a, const Box6& o) {
    for (int k = 0; k < 3; k++) { a.mn[k] = sel_min(a.mn[k], o.mn[k]); a.mx[k] = sel_max(a.mx[k], o.mx[k]); }
}
__device__ __forceinline__ float half_area(const Box6& b) {   // bvh.rs:14-17
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ float surface_area(const Box6& b) {   // aabb.rs:42-45
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// Block-wide fold of a box through LDS (ordered-integer atomics; the result is the same for every thread).
template <int BS>
__device__ __forceinline__ Box6 block_box(const Box6& mine, uint32_t* sh /* 6 words */) {
    if (threadIdx.x < 6) sh[threadIdx.x] = threadIdx.x < 3 ? ORD_POS_INF : ORD_NEG_INF;
    __syncthreads();
    // one atomic per wave and component: fold inside the wave first
    Box6 w = mine;
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; k++) {
            w.mn[k] = sel_min(w.mn[k], __shfl_xor(w.mn[k], off));
            w.mx[k] = sel_max(w.mx[k], __shfl_xor(w.mx[k], off));
        }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; k++) { atomicMin(&sh[k], f2ord(w.mn[k])); atomicMax(&sh[3 + k], f2ord(w.mx[k])); }
    __syncthreads();
    Box6 r;
    for (int k = 0; k < 3; k++) { r.mn[k] = ord2f(sh[k]); r.mx[k] = ord2f(sh[3 + k]); }
    __syncthreads();
    return r;
}
// Block-wide exclusive prefix count of a flag; returns the count before this thread, `total` for the block.
template <int BS>
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t* sh /* BS / 64 + 1 words */, uint32_t& total) {
    const unsigned long long m = __ballot(flag);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sh[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t base = 0, sum = 0;
    for (uint32_t w = 0; w < (uint32_t)(BS / 64); w++) {
        const uint32_t c = sh[w];
        base += w < wave ? c : 0u;
        sum += c;
    }
    __syncthreads();
    total = sum;
    return base + before;
}

// One node of the tree: bvh.rs:100-149 (construct_node) with partition_sah (bvh.rs:10-97) inlined.
template <int BS>
__global__ __launch_bounds__(BS) void k_bvh_level(const float* __restrict__ boxes, uint32_t* order, uint32_t* tmp_a, uint32_t* tmp_b, const BSeg* segs,
                                                  uint32_t n_segs, BNode* nodes, BSeg* next_small, BSeg* next_large, uint32_t* counters /* [0] nodes, [1] next small, [2] next large */,
                                                  float hit_cost) {
    __shared__ uint32_t sh_box[6];
    __shared__ uint32_t sh_rank[BS / 64 + 1];
    __shared__ uint32_t sh_count[3][16];
    __shared__ uint32_t sh_bin[3][16][6];
    __shared__ float sh_cost[48];
    __shared__ float sh_mid[48];
    __shared__ uint32_t sh_pick[4];
    for (uint32_t s = blockIdx.x; s < n_segs; s += gridDim.x) {
        const BSeg seg = segs[s];
        const uint32_t b = seg.begin, n = seg.end - seg.begin;
        // ---- bounds of the range and of its centroids (bvh.rs:104-113, :26-34) ----
        Box6 bb = empty_box(), cb = empty_box();
        for (uint32_t k = threadIdx.x; k < n; k += BS) {
            const Box6 x = load_box(boxes, order[b + k]);
            grow(bb, x);
            Box6 c;
            for (int a = 0; a < 3; a++) { const float v = (x.mn[a] + x.mx[a]) * 0.5f; c.mn[a] = v; c.mx[a] = v; }
            grow(cb, c);
        }
        bb = block_box<BS>(bb, sh_box);
        cb = block_box<BS>(cb, sh_box);
        // ---- 16 buckets per axis: count and box (bvh.rs:36-50) ----
        for (uint32_t q = threadIdx.x; q < 48; q += BS) {
            sh_count[q / 16][q % 16] = 0u;
            for (int c = 0; c < 6; c++) sh_bin[q / 16][q % 16][c] = c < 3 ? ORD_POS_INF : ORD_NEG_INF;
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += BS) {
            const Box6 x = load_box(boxes, order[b + k]);
            for (int a = 0; a < 3; a++) {
                const float c = (x.mn[a] + x.mx[a]) * 0.5f;
                const float ratio = (c - cb.mn[a]) / (cb.mx[a] - cb.mn[a]);
                const uint32_t bk = bucket_of(ratio);
                atomicAdd(&sh_count[a][bk], 1u);
                for (int q = 0; q < 3; q++) { atomicMin(&sh_bin[a][bk][q], f2ord(x.mn[q])); atomicMax(&sh_bin[a][bk][3 + q], f2ord(x.mx[q])); }
            }
        }
        __syncthreads();
        // ---- the 45 candidate planes (bvh.rs:52-72): thread (axis, i) prices plane i of its axis ----
        if (threadIdx.x < 48) {
            const uint32_t a = threadIdx.x / 16, i = threadIdx.x % 16;
            float cost = __builtin_huge_valf(), mid = 0.0f;
            bool valid = false;
            if (i >= 1) {
                uint32_t l = 0, r = 0;
                Box6 lb = empty_box(), rb = empty_box();
                for (uint32_t q = 0; q < 16; q++) {
                    Box6 x;
                    for (int c = 0; c < 3; c++) { x.mn[c] = ord2f(sh_bin[a][q][c]); x.mx[c] = ord2f(sh_bin[a][q][3 + c]); }
                    if (q < i) { l += sh_count[a][q]; grow(lb, x); }
                    else { r += sh_count[a][q]; grow(rb, x); }
                }
                if (l != 0 && r != 0) {
                    cost = (float)l * half_area(lb) + (float)r * half_area(rb);
                    mid = ((float)i / 16.0f) * (cb.mx[a] - cb.mn[a]) + cb.mn[a];
                    valid = true;
                }
            }
            sh_cost[threadIdx.x] = valid ? cost : __uint_as_float(0x7FC00000u);   // NaN: never below anything
            sh_mid[threadIdx.x] = mid;
        }
        __syncthreads();
        if (threadIdx.x == 0) {   // first strictly cheapest, axis major (bvh.rs:67)
            float best = __builtin_huge_valf();
            uint32_t pick = 48;
            for (uint32_t q = 0; q < 48; q++)
                if (sh_cost[q] < best) { best = sh_cost[q]; pick = q; }
            sh_pick[0] = pick;
        }
        __syncthreads();
        const uint32_t pick = sh_pick[0];
        const uint32_t axis = pick < 48 ? pick / 16 : 0u;
        const float mid = pick < 48 ? sh_mid[pick] : 0.0f;
        __syncthreads();
        // ---- in-place partition, centroid >= mid to the left (bvh.rs:74-94) ----
        auto goes_left = [&](uint32_t k) {
            const Box6 x = load_box(boxes, order[b + k]);
            return (x.mn[axis] + x.mx[axis]) * 0.5f >= mid;
        };
        uint32_t L = 0;
        for (uint32_t base = 0; base < n; base += BS) {
            const uint32_t k = base + threadIdx.x;
            uint32_t tot;
            block_rank<BS>(k < n && goes_left(k), sh_rank, tot);
            L += tot;
        }
        // misplaced elements of the left part in position order, of the right part from the far end
        uint32_t n_bad = 0;
        for (uint32_t base = 0; base < L; base += BS) {
            const uint32_t k = base + threadIdx.x;
            const bool bad = k < L && !goes_left(k);
            uint32_t tot;
            const uint32_t r = block_rank<BS>(bad, sh_rank, tot);
            if (bad) tmp_a[b + n_bad + r] = k;
            n_bad += tot;
        }
        uint32_t n_bad_r = 0;
        for (uint32_t base = 0; base < n - L; base += BS) {
            const uint32_t t = base + threadIdx.x;            // distance from the far end
            const bool in = t < n - L;
            const uint32_t k = in ? n - 1u - t : 0u;
            const bool bad = in && goes_left(k);
            uint32_t tot;
            const uint32_t r = block_rank<BS>(bad, sh_rank, tot);
            if (bad) tmp_b[b + n_bad_r + r] = k;
            n_bad_r += tot;
        }
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < n_bad; r += BS) {   // n_bad == n_bad_r
            const uint32_t i = b + tmp_a[b + r], j = b + tmp_b[b + r];
            const uint32_t x = order[i];
            order[i] = order[j];
            order[j] = x;
        }
        __syncthreads();
        // ---- split or leaf (bvh.rs:118-146) ----
        const uint32_t l = L, r = n - L;
        bool split = false;
        if (l > 0 && r > 0) {
            Box6 left = empty_box();
            for (uint32_t k = threadIdx.x; k < l; k += BS) grow(left, load_box(boxes, order[b + k]));
            left = block_box<BS>(left, sh_box);
            const float ls = surface_area(left);
            const float rs = surface_area(left);   // the reference folds the left range here too (bvh.rs:129-132)
            const float sa = surface_area(bb);
            const float split_cost = 1.0f + hit_cost * (ls / sa) * (float)l + hit_cost * (rs / sa) * (float)r;
            const float nosplit_cost = hit_cost * (float)(l + r);
            split = split_cost < nosplit_cost;
        }
        if (threadIdx.x == 0) {
            BNode nd;
            for (int k = 0; k < 3; k++) { nd.mn[k] = bb.mn[k]; nd.mx[k] = bb.mx[k]; }
            nd.begin = b; nd.left = NONE; nd.right = NONE; nd.size = 1; nd.pre = 0; nd.pad = 0;
            if (split) {
                const uint32_t child = atomicAdd(&counters[0], 2u);
                nd.left = child; nd.right = child + 1;
                const BSeg sl{b, b + l, child, 0u}, sr{b + l, b + n, child + 1, 0u};
                if (l <= SMALL_SEGMENT) next_small[atomicAdd(&counters[1], 1u)] = sl; else next_large[atomicAdd(&counters[2], 1u)] = sl;
                if (r <= SMALL_SEGMENT) next_small[atomicAdd(&counters[1], 1u)] = sr; else next_large[atomicAdd(&counters[2], 1u)] = sr;
            }
            nodes[seg.node] = nd;
        }
        __syncthreads();
    }
}

// Subtree sizes, deepest level first; pre-order indices, root first; then the reference's node records.
__global__ void k_bvh_sizes(BNode* nodes, uint32_t first, uint32_t last) {
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= last) return;
    BNode& nd = nodes[i];
    nd.size = nd.left == NONE ? 1u : 1u + nodes[nd.left].size + nodes[nd.right].size;
}
__global__ void k_bvh_preorder(BNode* nodes, uint32_t first, uint32_t last) {
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= last) return;
    const BNode nd = nodes[i];
    if (nd.left != NONE) {
        nodes[nd.left].pre = nd.pre + 1u;
        nodes[nd.right].pre = nd.pre + 1u + nodes[nd.left].size;
    }
}
__global__ void k_bvh_emit(const BNode* nodes, uint32_t n_nodes, uint32_t n_prims, rtamd_bvh_node* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes) {
        const BNode nd = nodes[i];
        rtamd_bvh_node o;
        for (int k = 0; k < 3; k++) { o.min[k] = nd.mn[k]; o.max[k] = nd.mx[k]; }
        o.skip = nd.pre + nd.size;   // bvh.rs:148: the node after the subtree
        o.first = nd.begin;
        out[nd.pre] = o;
    }
    if (i == 0) {   // sentinel (empty box, m, n), bvh.rs:164-165
        rtamd_bvh_node s;
        for (int k = 0; k < 3; k++) { s.min[k] = __builtin_huge_valf(); s.max[k] = -__builtin_huge_valf(); }
        s.skip = n_nodes; s.first = n_prims;
        out[n_nodes] = s;
    }
}
__global__ void k_bvh_iota(uint32_t* order, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = i;
}

}  // namespace

// boxes: n x 6 floats on the device; nodes_out: room for 2 n nodes (at most 2 n - 1 plus the sentinel); order_out: n.
hipError_t build_bvh_device(const float* boxes, uint32_t n, float hit_cost, rtamd_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* order_out, int num_cus,
                            hipStream_t st) {
#define BVH_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return e_; } } while (0)
    std::vector<void*> owned;
    auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); };
    auto alloc = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) owned.push_back(*p); return e; };
    BNode* nodes = nullptr;
    BSeg* q[4] = {nullptr, nullptr, nullptr, nullptr};   // small / large queues, double buffered
    uint32_t *tmp_a = nullptr, *tmp_b = nullptr, *counters = nullptr;
    BVH_TRY(alloc((void**)&nodes, (size_t)2 * n * sizeof(BNode)));
    for (int k = 0; k < 4; k++) BVH_TRY(alloc((void**)&q[k], ((size_t)n + 1) * sizeof(BSeg)));
    BVH_TRY(alloc((void**)&tmp_a, (size_t)n * 4));
    BVH_TRY(alloc((void**)&tmp_b, (size_t)n * 4));
    BVH_TRY(alloc((void**)&counters, 16));
    hipLaunchKernelGGL(k_bvh_iota, dim3((n + 255) / 256), dim3(256), 0, st, order_out, n);
    const BSeg root{0u, n, 0u, 0u};
    uint32_t h_counts[3] = {1u, 0u, 0u};   // nodes so far, small, large
    uint32_t n_small = n <= SMALL_SEGMENT ? 1u : 0u, n_large = 1u - n_small;
    BVH_TRY(hipMemcpyAsync(n_small ? q[0] : q[1], &root, sizeof(root), hipMemcpyHostToDevice, st));
    std::vector<uint32_t> level_first{0u};   // node index where each level starts
    int cur = 0;
    while (n_small + n_large > 0) {
        level_first.push_back(h_counts[0]);
        const uint32_t zero2[2] = {0u, 0u};
        BVH_TRY(hipMemcpyAsync(counters, &h_counts[0], 4, hipMemcpyHostToDevice, st));
        BVH_TRY(hipMemcpyAsync(counters + 1, zero2, 8, hipMemcpyHostToDevice, st));
        BSeg *small_in = q[2 * cur], *large_in = q[2 * cur + 1], *small_out = q[2 * (1 - cur)], *large_out = q[2 * (1 - cur) + 1];
        const uint32_t cap_blocks = (uint32_t)num_cus * 32u;
        if (n_small) hipLaunchKernelGGL(k_bvh_level<64>, dim3(n_small < cap_blocks ? n_small : cap_blocks), dim3(64), 0, st, boxes, order_out, tmp_a, tmp_b, small_in, n_small, nodes,
                                        small_out, large_out, counters, hit_cost);
        if (n_large) hipLaunchKernelGGL(k_bvh_level<1024>, dim3(n_large < cap_blocks ? n_large : cap_blocks), dim3(1024), 0, st, boxes, order_out, tmp_a, tmp_b, large_in, n_large,
                                        nodes, small_out, large_out, counters, hit_cost);
        BVH_TRY(hipGetLastError());
        BVH_TRY(hipMemcpyAsync(h_counts, counters, 12, hipMemcpyDeviceToHost, st));
        BVH_TRY(hipStreamSynchronize(st));
        n_small = h_counts[1]; n_large = h_counts[2];
        cur = 1 - cur;
    }
    const uint32_t n_nodes = h_counts[0];
    level_first.push_back(n_nodes);
    // level_first[k] .. level_first[k + 1]: nodes processed by level k - 1 ... the root is level_first[0] = 0 .. 1
    const size_t levels = level_first.size() - 1;
    for (size_t k = levels; k-- > 0;) {
        const uint32_t f = level_first[k], l = level_first[k + 1];
        if (l > f) hipLaunchKernelGGL(k_bvh_sizes, dim3((l - f + 255) / 256), dim3(256), 0, st, nodes, f, l);
    }
    for (size_t k = 0; k < levels; k++) {
        const uint32_t f = level_first[k], l = level_first[k + 1];
        if (l > f) hipLaunchKernelGGL(k_bvh_preorder, dim3((l - f + 255) / 256), dim3(256), 0, st, nodes, f, l);
    }
    hipLaunchKernelGGL(k_bvh_emit, dim3((n_nodes + 255) / 256), dim3(256), 0, st, nodes, n_nodes, n, nodes_out);
    BVH_TRY(hipGetLastError());
    BVH_TRY(hipStreamSynchronize(st));
    *n_nodes_out = n_nodes + 1;   // with the sentinel
    cleanup();
    return hipSuccess;
#undef BVH_TRY
}

}  // namespace rtamd
