// 8(f)-3: the shading cache resident in HBM and the BRDF trainer's shading combine.
//
// Reference: the loader concatenates the 13 baked maps of every view into a (pixels, 39) f32 table
// (utils/dataset/scannetpp/dataset.py:359-377), a training batch slices rows of it by a random pixel permutation (:409-414),
// and train_brdf_crf.py:195-203 combines one row with the material net's albedo / metallic / roughness:
//     kd = albedo*(1-metallic);  ks = 0.04*(1-metallic) + albedo*metallic
//     L  = kd*diffuse + ks*lerp_specular(specular0, roughness) + lerp_specular(specular1, roughness)
//
// Layout here (ours to choose, the values are the reference's): one row = 4 + 6R floats rounded up to a multiple of 4
// (160 B for R = 6, 16-B aligned)
//     [ d.r d.g d.b 0 | level 0: s0.rgb s1.rgb | level 1: s0.rgb s1.rgb | ... ]
// so the two roughness levels lerp_specular touches (floor, ceil) are ONE contiguous 48-B span: a random-row batch reads
// 16 + 48 B out of the 160-B row (one or two 128-B lines) instead of five scattered 12-B pieces of a 156-B unaligned row.
// All kernels are one thread per pixel, HBM-bound; no LDS (nothing is shared between pixels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iris {

constexpr int kMaxLevels = 8;
struct CacheMaps { const float* diffuse; const float* s0[kMaxLevels]; const float* s1[kMaxLevels]; };

__host__ __device__ inline int cache_row_floats(int R) { return (4 + 6 * R + 3) & ~3; }  // rows stay 16-B aligned for odd R too

// thread per output float4 (coalesced 16-B stores); the 12-B source pixels of neighbouring threads are neighbours too
__global__ void cache_pack_kernel(CacheMaps m, int64_t n, int R, float* __restrict__ rows) {
    const int RS = cache_row_floats(R), q4 = RS / 4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n * q4; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / q4;
        const int k0 = (int)(t - i * q4) * 4;
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            float v = 0.f;
            if (k < 3) v = m.diffuse[i * 3 + k];
            else if (k >= 4 && k < 4 + 6 * R) {
                const int kk = k - 4, j = kk / 6, c = kk - j * 6;
                v = c < 3 ? m.s0[j][i * 3 + c] : m.s1[j][i * 3 + c - 3];
            }
            rows[i * RS + k] = v;
        }
    }
}

// rows -> the reference's slice: out (B, 3+6R) = [diffuse | specular0 (R,3) | specular1 (R,3)]   (dataset.py:409-414)
__global__ void cache_gather_kernel(const float* __restrict__ rows, const int64_t* __restrict__ idx, int64_t B, int R, float* __restrict__ out) {
    const int S = 3 + 6 * R, RS = cache_row_floats(R);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < B * S; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / S;
        const int e = (int)(t - i * S);
        const float* q = rows + (idx ? idx[i] : i) * RS;
        int k;
        if (e < 3) k = e;
        else if (e < 3 + 3 * R) { const int j = (e - 3) / 3, c = (e - 3) - j * 3; k = 4 + 6 * j + c; }
        else { const int ee = e - 3 - 3 * R, j = ee / 3, c = ee - j * 3; k = 4 + 6 * j + 3 + c; }
        out[t] = q[k];
    }
}

struct LerpPos { int r0, r1; float w; };
__device__ __forceinline__ LerpPos lerp_position(float rough, int R) {  // utils/ops.py:108-115
    LerpPos p;
    const float r = (rough - 0.02f) / (float)(1.0 - 0.02) * (float)(R - 1);
    const float f = floorf(r);
    p.r1 = min(max((int)ceilf(r), 0), R - 1);
    p.r0 = min(max((int)f, 0), R - 1);
    p.w = r - f;
    return p;
}

struct CacheRow { float d[3], s0a[3], s1a[3], s0b[3], s1b[3]; };
__device__ __forceinline__ CacheRow load_row(const float* __restrict__ q, const LerpPos& p) {
    CacheRow r;
    const float4 d = *reinterpret_cast<const float4*>(q);
    r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z;
    const float2* a = reinterpret_cast<const float2*>(q + 4 + 6 * p.r0);   // (16 + 24 r0) B: 8-B aligned
    const float2 a0 = a[0], a1 = a[1], a2 = a[2];
    r.s0a[0] = a0.x; r.s0a[1] = a0.y; r.s0a[2] = a1.x; r.s1a[0] = a1.y; r.s1a[1] = a2.x; r.s1a[2] = a2.y;
    const float2* b = reinterpret_cast<const float2*>(q + 4 + 6 * p.r1);
    const float2 b0 = b[0], b1 = b[1], b2 = b[2];
    r.s0b[0] = b0.x; r.s0b[1] = b0.y; r.s0b[2] = b1.x; r.s1b[0] = b1.y; r.s1b[1] = b2.x; r.s1b[2] = b2.y;
    return r;
}

__global__ void shade_cached_fwd_kernel(const float* __restrict__ rows, const int64_t* __restrict__ idx, const float* __restrict__ albedo,
                                        const float* __restrict__ metallic, const float* __restrict__ roughness, int64_t B, int R,
                                        float* __restrict__ L) {
    const int RS = cache_row_floats(R);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const LerpPos p = lerp_position(roughness[i], R);
        const CacheRow r = load_row(rows + (idx ? idx[i] : i) * RS, p);
        const float m = metallic[i], m1 = 1.f - m, w1 = 1.f - p.w;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = albedo[i * 3 + c];
            const float kd = a * m1, ks = 0.04f * m1 + a * m;
            const float S0 = r.s0a[c] * w1 + r.s0b[c] * p.w;
            const float S1 = r.s1a[c] * w1 + r.s1b[c] * p.w;
            const float Ld = kd * r.d[c], Ls = ks * S0 + S1;
            L[i * 3 + c] = Ld + Ls;
        }
    }
}

// dL/d albedo (B,3), dL/d metallic (B), dL/d roughness (B); fixed summation order c = 0,1,2 (the oracle's)
__global__ void shade_cached_bwd_kernel(const float* __restrict__ rows, const int64_t* __restrict__ idx, const float* __restrict__ albedo,
                                        const float* __restrict__ metallic, const float* __restrict__ roughness, const float* __restrict__ gL,
                                        int64_t B, int R, float* __restrict__ g_albedo, float* __restrict__ g_metallic,
                                        float* __restrict__ g_roughness) {
    const int RS = cache_row_floats(R);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const LerpPos p = lerp_position(roughness[i], R);
        const CacheRow r = load_row(rows + (idx ? idx[i] : i) * RS, p);
        const float m = metallic[i], m1 = 1.f - m, w1 = 1.f - p.w;
        float g_m = 0.f, g_m1 = 0.f, g_w = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = albedo[i * 3 + c], g = gL[i * 3 + c];
            const float S0 = r.s0a[c] * w1 + r.s0b[c] * p.w;
            const float ks = 0.04f * m1 + a * m;
            const float g_kd = g * r.d[c], g_ks = g * S0;
            if (g_albedo) g_albedo[i * 3 + c] = g_kd * m1 + g_ks * m;
            g_m += g_ks * a;
            g_m1 += g_kd * a + g_ks * 0.04f;
            g_w += (g * ks) * (r.s0b[c] - r.s0a[c]) + g * (r.s1b[c] - r.s1a[c]);
        }
        if (g_metallic) g_metallic[i] = g_m - g_m1;
        if (g_roughness) g_roughness[i] = g_w * (float)(R - 1) / (float)(1.0 - 0.02);
    }
}

}  // namespace iris
