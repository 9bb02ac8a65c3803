// Shared declarations between the kernel file and the C-ABI implementation.
#pragma once
#include "rt_device.h"

namespace rtamd {

static constexpr int TRACE_BLOCK = 512;  // threads per block of the traversal kernels (8 waves)

struct RayBatch { const float *ox, *oy, *oz, *dx, *dy, *dz, *tmax; };
struct HitBatch { uint32_t *obj, *prim; float *t, *u, *v; };

struct CameraDev { f3 pos, dir, horizontal, vertical; uint32_t width, height; };

// Owned pixel set of one render call (see rtamd_region in rtamd.h).
struct RegionDev {
    uint32_t x0, y0, w, h;
    uint32_t tile_w, tile_h, tile_first, tile_stride;
    uint32_t tiles_x, n_tiles;
};

// Wavefront state of one chunk of `cap` pixel slots, structure-of-arrays in HBM.
// Vectors are stored component-major: component c of element i lives at [c * cap + i], so a wave
// reads 64 consecutive floats per component.
struct WaveBufs {
    uint32_t cap;
    // per pixel slot, persistent over the samples of the pixel
    uint64_t* rng_lo;     // SmallRng state, low / high half
    uint64_t* rng_hi;
    float* sum;           // 3*cap: sum of finished samples
    float* L;             // 3*cap: radiance of the sample in flight
    float* beta;          // 3*cap: path throughput
    // ray queues, double buffered by bounce parity
    uint32_t* q_slot[2];  // cap: pixel slot of the path
    float* q_ray[2];      // 6*cap: pos.xyz, dir.xyz
    // hit records of the current queue
    uint32_t* h_mat;      // cap: material index or MISS
    float* h_f;           // 8*cap: u, v, pos.xyz, norm.xyz
    // lit queue: paths whose hit has a non-black diffuse colour
    uint32_t* lit_slot;   // cap
    float* lit_f;         // 9*cap: p.xyz, beta.xyz, diffuse.xyz
    float* smp;           // 7*S*cap: per light sample dir.xyz, t_max, ill.xyz
    uint32_t* counts;     // [0],[1]: queue lengths, [2]: lit queue length
    uint64_t* stats;      // [0] closest-hit rays, [1] any-hit rays
};

struct LaunchCfg {
    int num_cus = 256;
    int trace_blocks_per_cu = 4;
    int use_lds = 1;
};

hipError_t launch_trace_closest(const DevScene& S, const LaunchCfg& cfg, const RayBatch& rays, const HitBatch& out, size_t n, hipStream_t st);
hipError_t launch_trace_any(const DevScene& S, const LaunchCfg& cfg, const RayBatch& rays, uint8_t* out, size_t n, hipStream_t st);
hipError_t launch_raygen(const WaveBufs& W, const RegionDev& R, const CameraDev& cam, uint32_t chunk_begin, uint32_t chunk_n,
                         uint32_t si, uint32_t sj, uint32_t aa, bool first, hipStream_t st);
hipError_t launch_mark_miss(const WaveBufs& W, size_t upper, hipStream_t st);
hipError_t launch_wave_trace(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, uint32_t cur, size_t upper, hipStream_t st);
hipError_t launch_wave_shade(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, uint32_t cur, int depth, int max_depth,
                             int shadow_rays, size_t upper, hipStream_t st);
hipError_t launch_wave_shadow(const DevScene& S, const LaunchCfg& cfg, const WaveBufs& W, int shadow_rays, size_t upper, hipStream_t st);
hipError_t launch_finalize(const WaveBufs& W, const RegionDev& R, uint32_t chunk_begin, uint32_t chunk_n, float n_samples, float* out, hipStream_t st);
hipError_t launch_quantize(const float* rgb, size_t n, uint8_t* out, hipStream_t st);

}  // namespace rtamd
