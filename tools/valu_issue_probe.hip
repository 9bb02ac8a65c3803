// Measures how many plain (non-packed) f32 VALU wave-instructions one SIMD of this GPU issues per cycle, as a function
// of the number of waves resident on the SIMD: the ceiling the traversal kernel's instruction stream is priced against.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue_probe tools/valu_issue_probe.hip && ./valu_issue_probe
//
// Every wave runs ITER x 64 independent-chain v_fma_f32 (8 chains of 8 per unrolled body, so a single wave is not
// bound by the dependent-instruction latency), plus a variant with v_pk_fma_f32 for comparison.  Blocks of 64 x W
// threads, one block per CU-quarter... simply: grid = CUs x 4 blocks of W waves would not pin waves to SIMDs, so a block
// holds 4 x W waves (the hardware deals a block's waves round-robin to the 4 SIMDs of its CU) and the grid has one
// block per CU.  Cycles come from s_memtime (shader clock) taken by each wave around its loop; the slowest wave of the
// block counts.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

template <bool PACKED>
__global__ void k_probe(float* sink, unsigned long long* cycles, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * 0.5f, b5 = a5 * 0.5f, b6 = a6 * 0.5f, b7 = a7 * 0.5f;
    const float m = 0.999f, c = 1e-4f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (PACKED) {
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(const double*)&b0), "v"(*(const double*)&b2));
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3" : "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(const double*)&b4), "v"(*(const double*)&b6));
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3" : "+v"(*(double*)&b0), "+v"(*(double*)&b2) : "v"(*(const double*)&a0), "v"(*(const double*)&a2));
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3" : "+v"(*(double*)&b4), "+v"(*(double*)&b6) : "v"(*(const double*)&a4), "v"(*(const double*)&a6));
            } else {
                asm volatile(
                    "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                    "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "v"(m), "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    const int iters = 20000;
    float* sink;
    unsigned long long* cyc;
    CHECK(hipMalloc(&sink, (size_t)cus * 1024 * 4));
    CHECK(hipMalloc(&cyc, (size_t)cus * 16 * 8));
    for (int packed = 0; packed < 2; packed++)
        for (int w = 1; w <= 4; w *= 2) {   // waves per SIMD (blocks are limited to 1024 threads = 4 per SIMD)
            const int threads = 64 * 4 * w;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0, 0));
                if (packed) hipLaunchKernelGGL(k_probe<true>, dim3(cus), dim3(threads), 0, 0, sink, cyc, iters);
                else hipLaunchKernelGGL(k_probe<false>, dim3(cus), dim3(threads), 0, 0, sink, cyc, iters);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipDeviceSynchronize());
            }
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h((size_t)cus * 4 * w);
            CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long mx = 0;
            double avg = 0;
            for (auto v : h) { mx = v > mx ? v : mx; avg += (double)v; }
            avg /= (double)h.size();
            // wave-instructions per wave: plain 64 per iteration, packed 8 x 8 = 64 too (each v_pk_fma is 2 flops-lanes wide)
            const double insts_per_wave = (double)iters * 64.0;
            const double per_simd = insts_per_wave * w;   // issued by one SIMD
            printf("%s  waves/SIMD %d: s_memtime ticks per wave avg %.0f max %llu | SIMD issues one wave-instruction per %.2f ticks | wall %.3f ms -> %.2f Ginst/s per SIMD-equivalent chip total %.1f Ginst/s\n",
                   packed ? "v_pk_fma_f32" : "v_fma_f32   ", w, avg, mx, (double)mx / per_simd, ms, per_simd / (ms * 1e6), per_simd * cus * 4 / (ms * 1e6));
        }
    return 0;
}
