// 8(f)-4: denoiser substitute for mitsuba.OptixDenoiser (bake_shading.py:81,129,198-200).
//
// The reference hands each baked map to the OptiX AI denoiser (closed, NVIDIA-only, no guides).  There is nothing to be bit-compatible
// with; the substitute is a variance-guided edge-avoiding a-trous wavelet filter (Dammertz et al. 2010; the spatial part of SVGF,
// Schied et al. 2017) that uses what this pipeline has and OptiX is not given: the primary-hit normal and position of every pixel.
//   1. guides   : g0 = (n.xyz, valid), g1 = (x.xyz, 0)  packed once per view (shared by all maps of the view)
//   2. variance : per map, luminance mean / variance in a 7x7 window weighted by the geometric weights  -> (rgb, var) float4
//   3. a-trous  : `iterations` passes, 5x5 B3-spline taps at stride 2^i, weight = h * w_n * w_p * w_l
//        w_n = max(0, n_p.n_q)^sigma_n                     w_p = exp(-|n_p.(x_q-x_p)| / (sigma_p*|x_q-x_p| + 1e-12))   (scale free)
//        w_l = exp(-|l_p-l_q| / (sigma_l*sqrt(gauss3x3(var)_p) + 1e-6));  colour' = sum w c / sum w;  var' = sum w^2 var / (sum w)^2
// Maps are processed M <= 4 at a time so that the taps' guide loads and geometric weights are shared (spec0 / spec1 of a roughness
// level see the same geometry).  Image-space stencils: L2/HBM-bound, one thread per pixel, 16x16 tiles so that a workgroup's taps
// overlap in L1/L2.  Invalid pixels (no primary hit) neither contribute nor receive (they stay 0, bake_shading.py:126-127).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iris {

constexpr int kDnMaxMaps = 4;
struct DnMaps { const float* in[kDnMaxMaps]; float* out[kDnMaxMaps]; float4* a[kDnMaxMaps]; float4* b[kDnMaxMaps]; };
struct DnParams { int H, W; float sigma_l, sigma_n, sigma_p; };

__device__ __forceinline__ float dn_lum(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }

// geometric weight between the centre (n_p, x_p) and a tap (g0q = n_q|valid, g1q = x_q)
__device__ __forceinline__ float dn_geo_weight(const float4& g0p, const float4& g1p, const float4& g0q, const float4& g1q, const DnParams& P) {
    if (g0q.w == 0.f) return 0.f;
    const float nn = fmaxf(0.f, g0p.x * g0q.x + g0p.y * g0q.y + g0p.z * g0q.z);
    const float wn = nn > 0.f ? exp2f(P.sigma_n * log2f(nn)) : 0.f;
    const float dx = g1q.x - g1p.x, dy = g1q.y - g1p.y, dz = g1q.z - g1p.z;
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    const float plane = fabsf(g0p.x * dx + g0p.y * dy + g0p.z * dz);
    const float wp = expf(-plane / (P.sigma_p * dist + 1e-12f));
    return wn * wp;
}

__global__ void dn_guides_kernel(const float* __restrict__ normal, const float* __restrict__ position, const uint8_t* __restrict__ valid, int64_t n,
                                 float4* __restrict__ g0, float4* __restrict__ g1) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool v = valid ? valid[i] != 0 : true;
        g0[i] = normal && v ? make_float4(normal[i * 3], normal[i * 3 + 1], normal[i * 3 + 2], 1.f) : make_float4(0.f, 0.f, 1.f, v ? 1.f : 0.f);
        g1[i] = position && v ? make_float4(position[i * 3], position[i * 3 + 1], position[i * 3 + 2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// 16x16 pixel tiles
__device__ __forceinline__ bool dn_pixel(const DnParams& P, int& x, int& y) {
    x = blockIdx.x * 16 + (threadIdx.x & 15);
    y = blockIdx.y * 16 + (threadIdx.x >> 4);
    return x < P.W && y < P.H;
}

template <int M>
__global__ __launch_bounds__(256) void dn_variance_kernel(DnParams P, DnMaps mp, const float4* __restrict__ g0, const float4* __restrict__ g1) {
    int x, y;
    if (!dn_pixel(P, x, y)) return;
    const int64_t p = (int64_t)y * P.W + x;
    const float4 g0p = g0[p], g1p = g1[p];
    if (g0p.w == 0.f) {
#pragma unroll
        for (int m = 0; m < M; ++m) mp.a[m][p] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float ws = 0.f, m1[M], m2[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { m1[m] = 0.f; m2[m] = 0.f; }
    for (int dy = -3; dy <= 3; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= P.H) continue;
        for (int dx = -3; dx <= 3; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= P.W) continue;
            const int64_t q = (int64_t)yy * P.W + xx;
            const float w = (dx == 0 && dy == 0) ? 1.f : dn_geo_weight(g0p, g1p, g0[q], g1[q], P);
            if (w == 0.f) continue;
            ws += w;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float l = dn_lum(mp.in[m][q * 3], mp.in[m][q * 3 + 1], mp.in[m][q * 3 + 2]);
                m1[m] += w * l; m2[m] += w * l * l;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const float mean = m1[m] / ws;
        const float var = fmaxf(0.f, m2[m] / ws - mean * mean);
        mp.a[m][p] = make_float4(mp.in[m][p * 3], mp.in[m][p * 3 + 1], mp.in[m][p * 3 + 2], var);
    }
}

// one a-trous pass a -> b at stride `step`; LAST writes the (H*W,3) output instead
template <int M, bool LAST>
__global__ __launch_bounds__(256) void dn_atrous_kernel(DnParams P, DnMaps mp, const float4* __restrict__ g0, const float4* __restrict__ g1, int step) {
    int x, y;
    if (!dn_pixel(P, x, y)) return;
    const int64_t p = (int64_t)y * P.W + x;
    const float4 g0p = g0[p], g1p = g1[p];
    if (g0p.w == 0.f) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (LAST) { mp.out[m][p * 3] = 0.f; mp.out[m][p * 3 + 1] = 0.f; mp.out[m][p * 3 + 2] = 0.f; }
            else mp.b[m][p] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    // 3x3 gaussian of the variance at the centre (valid pixels only)
    float inv_sl[M], lp[M];
    float4 cp[M];
    {
        float gw = 0.f, gv[M];
#pragma unroll
        for (int m = 0; m < M; ++m) gv[m] = 0.f;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx, yy = y + dy;
                if (xx < 0 || xx >= P.W || yy < 0 || yy >= P.H) continue;
                const int64_t q = (int64_t)yy * P.W + xx;
                if (g0[q].w == 0.f) continue;
                const float k = (dx == 0 ? 2.f : 1.f) * (dy == 0 ? 2.f : 1.f);
                gw += k;
#pragma unroll
                for (int m = 0; m < M; ++m) gv[m] += k * mp.a[m][q].w;
            }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            cp[m] = mp.a[m][p];
            lp[m] = dn_lum(cp[m].x, cp[m].y, cp[m].z);
            inv_sl[m] = 1.f / (P.sigma_l * sqrtf(fmaxf(0.f, gv[m] / gw)) + 1e-6f);
        }
    }
    float sr[M], sg[M], sb[M], sv[M], sw[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { sr[m] = cp[m].x; sg[m] = cp[m].y; sb[m] = cp[m].z; sv[m] = cp[m].w; sw[m] = 1.f; }  // centre tap: h = 1 (normalised below)
    const float h1[3] = {3.f / 8.f, 1.f / 4.f, 1.f / 16.f};
    const float hc = h1[0] * h1[0];
    for (int j = -2; j <= 2; ++j) {
        const int yy = y + j * step;
        if (yy < 0 || yy >= P.H) continue;
        for (int i = -2; i <= 2; ++i) {
            const int xx = x + i * step;
            if (xx < 0 || xx >= P.W || (i == 0 && j == 0)) continue;
            const int64_t q = (int64_t)yy * P.W + xx;
            const float wg = dn_geo_weight(g0p, g1p, g0[q], g1[q], P);
            if (wg == 0.f) continue;
            const float h = h1[i < 0 ? -i : i] * h1[j < 0 ? -j : j] / hc * wg;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float4 c = mp.a[m][q];
                const float w = h * expf(-fabsf(lp[m] - dn_lum(c.x, c.y, c.z)) * inv_sl[m]);
                sr[m] += w * c.x; sg[m] += w * c.y; sb[m] += w * c.z; sv[m] += w * w * c.w; sw[m] += w;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const float inv = 1.f / sw[m];
        if (LAST) { mp.out[m][p * 3] = sr[m] * inv; mp.out[m][p * 3 + 1] = sg[m] * inv; mp.out[m][p * 3 + 2] = sb[m] * inv; }
        else mp.b[m][p] = make_float4(sr[m] * inv, sg[m] * inv, sb[m] * inv, sv[m] * inv * inv);
    }
}

}  // namespace iris
