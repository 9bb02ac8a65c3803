// Per-instruction cost table of one gfx950 SIMD as a wave sees it: how many shader-clock ticks (s_memtime) a wave spends
// per instruction of a DEPENDENT chain of each kind the traversal scheduler's loop is made of, with 1, 2 and 4 waves on
// the SIMD.  The scheduler's waves are serial chains (vote -> LDS round trip -> arithmetic -> vote ...), so these numbers,
// not the issue peak, price a loop trip.
//
//   hipcc --offload-arch=gfx950 -O2 -o latency_table_probe tools/latency_table_probe.hip && ./latency_table_probe
//
// One block of 256 x W threads on one CU (its waves are dealt round-robin to the 4 SIMDs: W per SIMD); every wave runs
// ITERS x 16 copies of the op and lane 0 of each wave reports its ticks; the slowest wave counts.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

static constexpr int ITERS = 2000;
typedef __attribute__((address_space(3))) unsigned lds_u32;

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum Kind {
    K_FMA_DEP, K_FMA_2CHAIN, K_FMA_4CHAIN, K_MOV_DEP, K_SALU_DEP, K_IF_REGION, K_BRANCH_TAKEN, K_LDS_CHASE, K_LDS_B128, K_BPERMUTE,
    K_READLANE_ROUND, K_BALLOT_ROUND, K_DPP_STEP, K_GLOBAL_CHASE, K_SLOAD_CHASE, K_LDS_ATOMIC_READBACK, K_DIV, K_COUNT
};
static const char* kind_name[K_COUNT] = {
    "v_fma_f32, one dependent chain", "v_fma_f32, 2 chains (per instruction)", "v_fma_f32, 4 chains (per instruction)", "v_mov_b32 dependent chain",
    "s_add_u32 dependent chain", "if-region: v_cmp + s_and_saveexec + s_cbranch_execz (not taken) + s_or exec", "s_branch (taken, to the next instruction)",
    "ds_read_b32 pointer chase (+ s_waitcnt)", "ds_read_b128 + s_waitcnt + dependent VALU", "ds_bpermute_b32 + s_waitcnt (dependent)",
    "v_readfirstlane -> s_add -> v_mov (VALU->SALU->VALU round)", "v_cmp -> s_bcnt1 -> s_cmp -> s_cselect -> v_mov (ballot round)",
    "v_add_u32 dpp row_shr:1 + s_nop 1 (one scan step)", "global_load_dword pointer chase (L2 hit, + s_waitcnt)", "s_load_dword pointer chase (+ s_waitcnt)",
    "ds_min_u64 + ds_read_b64 read-back + s_waitcnt", "IEEE f32 division 1/x (dependent chain)"};

template <int KIND>
__global__ void k_probe(unsigned* gchain, float* sink, unsigned long long* ticks) {
    __shared__ unsigned lds[4096];
    for (unsigned i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (i * 4u + 256u * 4u) & (4096u * 4u - 1u);   // byte address of the next element
    __syncthreads();
    float a = threadIdx.x * 1e-3f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
    const float m = 0.999f, k = 1e-4f;
    unsigned u = KIND == K_LDS_B128 ? (threadIdx.x * 16u) & 0x3ff0u : threadIdx.x * 4u, s = 0;
    unsigned long long g = (unsigned long long)gchain + (unsigned long long)(threadIdx.x & 63) * 8ull;
    unsigned long long sp = (unsigned long long)gchain;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        if (KIND == K_FMA_DEP) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));) }
        if (KIND == K_FMA_2CHAIN) { REP4(REP4(asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(m), "v"(k));)) }
        if (KIND == K_FMA_4CHAIN) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(k));) }
        if (KIND == K_MOV_DEP) { REP16(asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %0" : "+v"(a), "+v"(b));) }
        if (KIND == K_SALU_DEP) { REP16(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");) }
        if (KIND == K_IF_REGION) {
            REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_saveexec_b64 s[8:9], vcc\n\ts_cbranch_execz 1f\n\tv_add_u32 %0, %0, 1\n1:\n\ts_or_b64 exec, exec, s[8:9]" : "+v"(u) : "v"(0x7FFFFFFFu) : "vcc", "s8", "s9", "scc");)
        }
        if (KIND == K_BRANCH_TAKEN) { REP16(asm volatile("s_branch 1f\n\ts_nop 0\n1:" ::: "memory");) }
        if (KIND == K_LDS_CHASE) { REP16(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(u)::"memory");) }
        if (KIND == K_LDS_B128) {
            REP16(asm volatile("ds_read_b128 v[20:23], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0x3ff0, v20" : "+v"(u)::"memory", "v20", "v21", "v22", "v23");)
        }
        if (KIND == K_BPERMUTE) { REP16(asm volatile("ds_bpermute_b32 %0, %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xfc, %0" : "+v"(u)::"memory");) }
        if (KIND == K_READLANE_ROUND) { REP16(asm volatile("v_readfirstlane_b32 s8, %0\n\ts_add_u32 s8, s8, 1\n\tv_mov_b32 %0, s8" : "+v"(u)::"s8", "scc");) }
        if (KIND == K_BALLOT_ROUND) {
            REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_bcnt1_i32_b64 s8, vcc\n\ts_cmp_lg_u32 s8, 0\n\ts_cselect_b32 s8, 1, 2\n\tv_add_u32 %0, s8, %0" : "+v"(u) : "v"(0x7FFFFFFFu) : "vcc", "s8", "scc");)
        }
        if (KIND == K_DPP_STEP) { REP16(asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u));) }
        if (KIND == K_GLOBAL_CHASE) { REP16(asm volatile("global_load_dwordx2 %0, %0, off\n\ts_waitcnt vmcnt(0)" : "+v"(g)::"memory");) }
        if (KIND == K_SLOAD_CHASE) { REP16(asm volatile("s_load_dwordx2 %0, %0, 0x0\n\ts_waitcnt lgkmcnt(0)" : "+s"(sp)::"memory");) }
        if (KIND == K_LDS_ATOMIC_READBACK) {
            REP16(asm volatile("ds_min_u64 %0, v[20:21]\n\tds_read_b64 v[20:21], %0\n\ts_waitcnt lgkmcnt(0)" : : "v"(u & 0x3ff8u) : "memory", "v20", "v21");)
        }
        if (KIND == K_DIV) { REP16(a = 1.0f / a; asm volatile("" : "+v"(a));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = a + b + c + d + (float)u + (float)s + (float)(unsigned)g + (float)(unsigned)sp;
}

template <int KIND>
int run_kind(unsigned* gchain, float* sink, unsigned long long* ticks) {
    if constexpr (KIND < K_COUNT) {
        const int per_iter[K_COUNT] = {16, 32, 64, 32, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
        printf("%-82s", kind_name[KIND]);
        for (int w = 1; w <= 4; w *= 2) {
            hipLaunchKernelGGL(k_probe<KIND>, dim3(1), dim3(256 * w), 0, 0, gchain, sink, ticks);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(4 * w);
            CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long mx = 0;
            for (auto v : h) mx = v > mx ? v : mx;
            printf("  %8.1f", (double)mx / ((double)ITERS * per_iter[KIND]));
        }
        printf("\n");
        return run_kind<KIND + 1>(gchain, sink, ticks);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("%s: ticks (s_memtime) per op seen by one wave, with 1 / 2 / 4 waves per SIMD running the same chain\n", prop.gcnArchName);
    unsigned* gchain;
    float* sink;
    unsigned long long* ticks;
    CHECK(hipMalloc(&gchain, 4096));
    CHECK(hipMalloc(&sink, 1024 * 4));
    CHECK(hipMalloc(&ticks, 16 * 8));
    // every 8-byte element of the chain points at itself: the chase stays on one cache line per lane
    std::vector<unsigned long long> self(512);
    for (int i = 0; i < 512; i++) self[i] = (unsigned long long)gchain + (unsigned long long)i * 8ull;
    CHECK(hipMemcpy(gchain, self.data(), 4096, hipMemcpyHostToDevice));
    return run_kind<0>(gchain, sink, ticks);
}
