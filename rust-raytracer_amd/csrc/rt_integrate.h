// Device functions of the integrator shared by the kernel files: the random stream, textures, materials, light sampling,
// the pixel-slot mapping and the camera rays.
#pragma once
#include "rt_device.h"
#include "rt_kernels.h"

namespace rtamd {

// ------------------------------------------------------------------------------------------------
// rand 0.6 SmallRng on 64-bit targets = PCG Mcg128Xsl64, seeded through rand_core's
// seed_from_u64 (PCG32 expansion).  State is the 128-bit MCG state as two 64-bit halves.
// ------------------------------------------------------------------------------------------------
struct Rng { uint64_t lo, hi; };
__device__ __forceinline__ Rng rng_seed(uint64_t state) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        state = state * MUL + INC;
        uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        w[k] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    Rng r;
    r.lo = ((uint64_t)w[0] | ((uint64_t)w[1] << 32)) | 1ULL;
    r.hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    return r;
}
__device__ __forceinline__ uint32_t rng_u32(Rng& r) {
    const uint64_t MH = 0x2360ED051FC65DA4ULL, ML = 0x4385DF649FCCF645ULL;
    uint64_t lo = r.lo * ML;
    uint64_t hi = __umul64hi(r.lo, ML) + r.lo * MH + r.hi * ML;
    r.lo = lo; r.hi = hi;
    uint32_t rot = (uint32_t)(hi >> 58);
    uint64_t xsl = hi ^ lo;
    uint64_t o = (xsl >> rot) | (xsl << ((64 - rot) & 63));
    return (uint32_t)o;
}
__device__ __forceinline__ float rng_f32(Rng& r) {  // Standard: 24 bits * 2^-24
    return (1.0f / 16777216.0f) * (float)(rng_u32(r) >> 8);
}

// ------------------------------------------------------------------------------------------------
// Shading helpers (src/material.rs, src/light.rs, src/scene.rs:16-21, src/texture.rs:70-82).
// ------------------------------------------------------------------------------------------------
static constexpr float PI_F = 3.14159274101257324219f;

__device__ __forceinline__ float rust_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
__device__ __forceinline__ float rust_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }

// Image::color_at, src/texture.rs:70-82: nearest texel, clamped.
__device__ __forceinline__ f3 image_texel(const DevScene& S, uint32_t first, uint32_t w, uint32_t h, float u, float v) {
    float fw = (float)w, fh = (float)h;
    float fx = rust_max(rust_min(u * fw, fw - 1.0f), 0.0f);
    float fy = rust_max(rust_min(v * fh, fh - 1.0f), 0.0f);
    uint32_t x = (uint32_t)fx, y = (uint32_t)fy;
    uint32_t p = S.sky[first + (size_t)y * w + x];
    return mk((float)(p & 0xFFu) / 255.0f, (float)((p >> 8) & 0xFFu) / 255.0f, (float)((p >> 16) & 0xFFu) / 255.0f);
}
__device__ __forceinline__ f3 sky_color(const DevScene& S, f3 d) {   // scene.rs:16-21
    float u = atan2f(d.x, d.y) / PI_F * 0.5f + 0.5f;
    float v = acosf(d.z) / PI_F;
    return image_texel(S, 0u, S.sky_w, S.sky_h, u, v);
}
__device__ __forceinline__ int32_t sat_i32(float x) {  // `as i32`: saturating, NaN -> 0
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return -2147483647 - 1;
    return (int32_t)x;
}
__device__ __forceinline__ f3 texture_color(const DevScene& S, const DevMaterial& m, float u, float v) {
    if (m.tex_kind == TEX_UV) return mk(u, v, 0.0f);                 // texture.rs:22-26
    if (m.tex_kind == TEX_CHECKERBOARD) {                             // texture.rs:36-47
        const int32_t i = sat_i32(u * (float)m.tex_param);
        const int32_t j = sat_i32(v * (float)m.tex_param);
        return ((i % 2 == 0) != (j % 2 == 0)) ? mk(0.f, 0.f, 0.f) : mk(1.f, 1.f, 1.f);
    }
    if (m.tex_kind == TEX_IMAGE) {
        const DevImage im = S.images[m.tex_param];
        return image_texel(S, im.first, im.w, im.h, u, v);
    }
    return mk(m.color[0], m.color[1], m.color[2]);                    // texture.rs:14-18
}

__device__ __forceinline__ f3 reflection(f3 n, f3 i) { return i - (2.0f * n) * dot(n, i); }  // material.rs:83-85
__device__ __forceinline__ bool refraction(f3 normal, f3 i, float ior, f3& o) {             // material.rs:87-101
    float side = dot(normal, i);
    float cosi, eta;
    f3 n;
    if (side < 0.0f) { cosi = -side; eta = 1.0f / ior; n = normal; }
    else { cosi = side; eta = ior; n = -normal; }
    float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
    if (k > 0.0f) {
        o = eta * i + (eta * cosi - sqrtf(k)) * n;
        return true;
    }
    return false;
}
__device__ __forceinline__ float fresnel(f3 normal, f3 i, float ior) {                      // material.rs:103-121
    float side = dot(normal, i);
    float cosi, etat, etai;
    if (side < 0.0f) { cosi = -side; etat = ior; etai = 1.0f; }
    else { cosi = side; etat = 1.0f; etai = ior; }
    float sint = etai / etat * sqrtf(1.0f - cosi * cosi);
    if (sint < 1.0f) {
        float cost = sqrtf(1.0f - sint * sint);
        float rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
        float rp = ((etat * cosi) - (etai * cost)) / ((etai * cosi) + (etat * cost));
        return (rs * rs + rp * rp) / 2.0f;
    }
    return 1.0f;
}
__device__ __forceinline__ f3 material_diffuse(const DevScene& S, const DevMaterial& m, float u, float v) {  // Material::sample_at
    if (m.kind == MAT_NULL) return mk(0.f, 1.f, 0.f);
    if (m.kind == MAT_LAMBERT) return texture_color(S, m, u, v);
    if (m.kind == MAT_GLOSSY) return texture_color(S, m, u, v) * (1.0f - m.p1);
    return mk(0.f, 0.f, 0.f);
}
__device__ __forceinline__ bool material_scatter(const DevMaterial& m, f3 n, f3 i, Rng& rng, f3& out_dir, f3& w) {
    if (m.kind == MAT_METAL) {  // material.rs:32-35
        out_dir = i - (2.0f * n) * dot(n, i);
        w = fill(1.0f);
        return true;
    }
    if (m.kind == MAT_GLOSSY) {  // material.rs:45-69 (accept test reads the mirror direction)
        f3 out = reflection(n, i);
        f3 a, b;
        ortho_axes(out, a, b);
        float side = dot(n, i) < 0.0f ? 1.0f : -1.0f;
        for (int k = 0; k < 10; k++) {
            float theta = rng_f32(rng) * 2.0f * PI_F;
            float u = rng_f32(rng);
            float p = powf(u, 2.0f / (1.0f + m.p0));
            float r = sqrtf(1.0f - p);
            float x = r * cosf(theta);
            float y = r * sinf(theta);
            float z = sqrtf(p);
            f3 o = (x * a + y * b) + z * out;
            if (dot(n, out) * side > 0.0f) {
                out_dir = o;
                w = fill(1.0f) * m.p1;
                return true;
            }
        }
        return false;
    }
    if (m.kind == MAT_TRANSPARENT) {  // material.rs:123-133
        f3 o;
        if (rng_f32(rng) < fresnel(n, i, m.p0)) {
            o = reflection(n, i);
        } else if (!refraction(n, i, m.p0, o)) {
            return false;
        }
        out_dir = o;
        w = fill(1.0f);
        return true;
    }
    return false;
}
__device__ __forceinline__ bool light_is_delta(const DevLight& l) {  // light.rs:29-31,76-78,125-127
    if (l.kind == LIGHT_AMBIENT) return true;
    if (l.kind == LIGHT_POINT) return l.param == 0.0f;
    if (l.kind == LIGHT_DIRECTION) return l.param > 0.0f;
    return false;
}
__device__ __forceinline__ void light_sample(const DevLight& l, f3 pos, f3 normal, Rng& rng, f3& dir, float& t_max, f3& ill) {
    f3 E = mk(l.emission[0], l.emission[1], l.emission[2]);
    if (l.kind == LIGHT_AMBIENT) {  // light.rs:24-27
        dir = normal; t_max = 0.0f; ill = E;
    } else if (l.kind == LIGHT_POINT) {  // light.rs:50-74
        // rejection-sample a point of the unit ball; gen_range(-1, 1) draws a float in [1, 2) from the top 23 bits of a
        // u32, maps it with value * scale + (low - scale), and redraws a result that rounds up to the high end
        f3 fv;
        for (;;) {
            float f[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                for (;;) {
                    const float v12 = __uint_as_float((rng_u32(rng) >> 9) | 0x3F800000u);
                    const float scale = 1.0f - (-1.0f);
                    const float res = v12 * scale + (-1.0f - scale);
                    if (res < 1.0f) { f[k] = res; break; }
                }
            }
            fv = mk(f[0], f[1], f[2]);
            if (dot(fv, fv) < 1.0f) break;
        }
        const f3 offset = (mk(l.vec[0], l.vec[1], l.vec[2]) + fv * l.param) - pos;
        const float dist_sq = dot(offset, offset);
        const float dist = sqrtf(dist_sq);
        dir = offset / dist;
        const float c = rust_max(dot(dir, normal), 0.0f);
        t_max = dist; ill = (E / dist_sq) * c;
    } else if (l.kind == LIGHT_DIRECTION) {  // light.rs:97-123
        f3 ld = mk(l.vec[0], l.vec[1], l.vec[2]);
        f3 o;
        if (l.param > 0.0f) {
            float theta = rng_f32(rng) * 2.0f * PI_F;
            float u = rng_f32(rng);
            float p = powf(u, l.param);
            float r = sqrtf(1.0f - p);
            float x = r * cosf(theta);
            float y = r * sinf(theta);
            float z = sqrtf(p);
            f3 a, b;
            ortho_axes(ld, a, b);
            o = (x * a + y * b) - z * ld;
        } else {
            o = -ld;
        }
        float c = rust_max(dot(o, normal), 0.0f);
        dir = o; t_max = 1e12f; ill = E * c;
    } else {  // LIGHT_AO, light.rs:144-158
        float theta = rng_f32(rng) * 2.0f * PI_F;
        float u = rng_f32(rng);
        float r = sqrtf(u);
        float x = r * cosf(theta);
        float y = r * sinf(theta);
        float z = sqrtf(1.0f - u);
        f3 a, b;
        ortho_axes(normal, a, b);
        dir = (a * x + b * y) + normal * z;
        t_max = l.param; ill = E;
    }
}

// Whether a sample of the light can need a shadow ray at all (illumination() skips the occlusion test where the
// light's t_max is 0, integrator.rs:124): an ambient light never does, an occlusion light of reach 0 neither.
__device__ __forceinline__ bool light_shoots(const DevLight& l) {
    if (l.kind == LIGHT_AMBIENT) return false;
    if (l.kind == LIGHT_AO) return !(l.param == 0.0f);
    return true;
}

// Order of the pixel slots inside a tile (RegionDev::squares, chosen by the host per frame): squares of 8 x 8 pixels, row of
// squares by row of squares (the 64 consecutive slots of a wave are then a square of the image and not a line of it: the
// texels, hit records and triangles its lanes read lie closer together), or the rows of the tile.
__device__ __forceinline__ void tile_xy(const RegionDev& R, uint32_t r, uint32_t& tx, uint32_t& ty) {
    if (R.squares != 0u) {
        const uint32_t b = r >> 6, in = r & 63u, per_row = R.tile_w >> 3;
        const uint32_t by = b / per_row, bx = b - by * per_row;
        tx = bx * 8u + (in & 7u); ty = by * 8u + (in >> 3);
    } else {
        ty = r / R.tile_w; tx = r - ty * R.tile_w;
    }
}

// Pixel slot -> pixel coordinates of the owned tile set (see rtamd_region).
__device__ __forceinline__ bool slot_pixel(const RegionDev& R, uint32_t slot, uint32_t& cx, uint32_t& cy) {
    uint32_t per_tile = R.tile_w * R.tile_h;
    uint32_t q = slot / per_tile, r = slot - q * per_tile;
    uint32_t tx, ty;
    tile_xy(R, r, tx, ty);
    uint32_t tile = R.tile_first + q * R.tile_stride;
    uint32_t tyi = tile / R.tiles_x, txi = tile - tyi * R.tiles_x;
    uint32_t px = txi * R.tile_w + tx, py = tyi * R.tile_h + ty;
    cx = R.x0 + px; cy = R.y0 + py;
    return px < R.w && py < R.h && tile < R.n_tiles;
}

// Pixel slot -> index of its pixel in the output image (owned tiles one after the other, each row-major: rtamd_region).
__device__ __forceinline__ size_t slot_out(const RegionDev& R, uint32_t slot) {
    uint32_t per_tile = R.tile_w * R.tile_h;
    uint32_t q = slot / per_tile, r = slot - q * per_tile;
    uint32_t tx, ty;
    tile_xy(R, r, tx, ty);
    return (size_t)q * per_tile + (size_t)ty * R.tile_w + tx;
}

// Direction of the camera ray through sample (si, sj) of pixel (cx, cy): integrator.rs:41-42, scene.rs:84-90.
__device__ __forceinline__ f3 primary_dir(const CameraDev& cam, uint32_t cx, uint32_t cy, uint32_t si, uint32_t sj, uint32_t aa) {
    float n = (float)aa;
    float x = (float)cx + ((float)si + 0.5f) / n - 0.5f;
    float y = (float)cy + ((float)sj + 0.5f) / n - 0.5f;
    float u = 2.0f * (x / (float)cam.width) - 1.0f;
    float v = 2.0f * (y / (float)cam.height) - 1.0f;
    f3 d = (cam.dir + u * cam.horizontal) + v * cam.vertical;
    return normalize(d);
}

}  // namespace rtamd
