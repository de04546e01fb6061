// Device-side building blocks of libiris_hip.so (gfx950 only): shading math, RNG, SLF / emitter lookup.
// Every function cites the reference code it implements.  Compiled with -ffp-contract=off; every fused
// multiply-add below is an explicit fmaf().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iris {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, f3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }

// ---- intersection arithmetic contract (must match oracle/iris_oracle.c bit for bit) ----
__device__ __forceinline__ f3 x_cross(f3 a, f3 b) {
    return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ float x_dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }

// ---- torch-order helpers (sum over the last dim in index order, no fma) ----
__device__ __forceinline__ float t_dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// (Three divisions by one denominator.  hipcc expands each into v_div_scale x 2, v_rcp_f32, two Newton steps, q = a r, two residual corrections, v_div_fmas,
//  v_div_fixup; for operands inside the normal range the scale / fixup steps are the identity, so the reciprocal's refinement can be SHARED by the three
//  numerators and the result stays bit-identical -- 18 instead of 33 instructions per normalisation, verified bit for bit at BASELINE configs[1] size.  Measured
//  -1.0 % on the view kernel (EXPERIMENTS.md round 4): fewer instructions, more live registers through the sampling code.  Not kept.)
// NF.normalize(v, dim=-1) = v / max(||v||, 1e-12)
__device__ __forceinline__ f3 t_normalize(f3 a) {
    float n = sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
    n = fmaxf(n, 1e-12f);
    return mk3(a.x / n, a.y / n, a.z / n);
}
__device__ __forceinline__ f3 t_cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float relu(float x) { return x > 0.f ? x : 0.f; }

constexpr float kPi = 3.14159265358979323846f;     // float32(math.pi)
constexpr float kTwoPi = 6.28318530717958647692f;  // float32(2*math.pi)
constexpr float kRayEps = 8.940696716308594e-05f;  // mitsuba.math.RayEpsilon (f32) = 1500*2^-24

// utils/ops.py:12-30 get_normal_space: tangent/bitangent for a unit normal (columns t,b,n).
__device__ __forceinline__ void normal_space(f3 n, f3& t, f3& b) {
    // mask = |dot((1,0,0),n)| <= 1e-1 (compared in f32): cross((1,0,0),n) = (0,-nz,ny); else cross((0,1,0),n) = (nz,0,-nx)
    if (fabsf(n.x) <= 0.1f) t = t_normalize(mk3(0.f, -n.z, n.y));
    else                    t = t_normalize(mk3(n.z, 0.f, -n.x));
    b = t_cross(n, t);
}
// (w[:,None] @ Nmat.permute(0,2,1)).squeeze(1)
__device__ __forceinline__ f3 to_world(f3 l, f3 t, f3 b, f3 n) {
    return mk3((l.x * t.x + l.y * b.x) + l.z * n.x, (l.x * t.y + l.y * b.y) + l.z * n.y, (l.x * t.z + l.y * b.z) + l.z * n.z);
}
// sin/cos of x in [0, 2*pi] with a fully specified IEEE operation sequence, so that the CPU oracle's "device arithmetic" mode reproduces it bit
// for bit: quadrant j = int(x * 2/pi + 0.5), z = x - j * pi/2 in DOUBLE (one fma), the two kernels as double Horner chains (explicit fma), one
// rounding to f32.  Double because that is the cheap way to a (practically) correctly rounded result on this chip -- v_fma_f64 issues at the rate of
// the 4-cycle f32 class, and the f32-only Cody-Waite version of rounds 1-3 (~1 ulp) accounted for 14 of the 15 sample flips the HIP path had
// on top of the libm restatement's against the reference (profiles/r4_flip_attribution.json).  Exhaustively over the 2^24 Philox outputs, phi =
// 2 pi u: equal to the correctly rounded sin / cos for all but 2e-5 of the inputs; equal to torch-CPU's for 95.1 % (the correctly rounded value: 95.1 %,
// glibc's sinf / cosf: 95.0 %; the old sequence: 80-85 %).
// FIRST_QUADRANT: the caller guarantees x <= pi/2 + a few ulps (a polar angle): the same values from two selects instead of six.
// (a double constant behind an optimisation barrier: it is materialised where it is used -- two scalar moves -- instead of being hoisted out of the
//  sampling loop into a register pair that is then spilled to scratch and reloaded per sample)
__device__ __forceinline__ double kd(double c) {
    asm volatile("" : "+s"(c));
    return c;
}
template <bool FIRST_QUADRANT = false>
__device__ __forceinline__ void spec_sincos(float x, float& s, float& c) {
    const int j = (int)(x * 0.636619772367581343f + 0.5f);
    const double z = fma(-(double)j, kd(1.57079632679489661923), (double)x);     // |z| <= pi/4 (+ rounding of the quadrant choice)
    const double zz = z * z;
    double p = kd(2.7249902524065394e-06);                                         // (sin z / z - 1) / z^2, |error| < 3e-11 on the interval
    p = fma(p, zz, kd(-0.0001984008661425884)); p = fma(p, zz, kd(0.00833333187464819)); p = fma(p, zz, kd(-0.16666666663855825));
    const float ps = (float)fma(z * zz, p, z);
    double q = kd(-2.723710465738025e-07);                                         // (cos z - 1) / z^2, |error| < 4e-13
    q = fma(q, zz, kd(2.4799861845569796e-05)); q = fma(q, zz, kd(-0.0013888885090442048)); q = fma(q, zz, kd(0.04166666663738883)); q = fma(q, zz, kd(-0.4999999999996389));
    const float pc = (float)fma(zz, q, 1.0);
    if (FIRST_QUADRANT) {   // j is 0 or 1
        s = j == 0 ? ps : pc;
        c = j == 0 ? pc : -ps;
        return;
    }
    const int k = j & 3;
    s = (k == 0) ? ps : (k == 1) ? pc : (k == 2) ? -ps : -pc;
    c = (k == 0) ? pc : (k == 1) ? -ps : (k == 2) ? -pc : ps;
}
// asin / acos on [0, 1] with a fully specified IEEE operation sequence (explicit fmaf, IEEE sqrtf, an integer-seeded Newton reciprocal):
// the reference evaluates sin / cos of the ROUNDED polar angle theta = asin(sqrt(u0)) / acos(sqrt(c2)) (model/brdf.py:28, :50-51), and near grazing
// cos(theta) inherits theta's rounding (6e-8 absolute) -- a closed form of cos(asin s) does not, and the sampled direction then differs from the
// reference's in every bit below that.  So theta itself is computed, as accurately as f32 allows: |x| < 0.5: x + x z P(z), z = x^2; otherwise
// pi/2 - 2 asin(sqrt z), z = (1 - x) / 2, with the square root's rounding error carried (r + rl) and the subtraction from pi/2 compensated
// (hi / lo parts, Fast2Sum).  Exhaustively over the 2^24 Philox outputs u0: equal to the correctly rounded asin(sqrt u0) for 98.5 % of the inputs,
// never more than 1 ulp off; equal to torch-CPU's asin for 94.1 % (the correctly rounded value: 94.9 %, glibc's asinf: 90.4 %).
__device__ __forceinline__ float spec_asin_poly(float z) {   // (asin(sqrt z) / sqrt z - 1) / z on [0, 0.25], |error| < 4e-9
    float p = 0.033805747f;
    p = fmaf(p, z, 0.01707786f); p = fmaf(p, z, 0.031116156f); p = fmaf(p, z, 0.04459803f); p = fmaf(p, z, 0.07500099f);
    return fmaf(p, z, 0.16666666f);
}
__device__ __forceinline__ float spec_half_rcp(float r) {    // 0.5 / r to ~2e-4 relative (scales a 6e-8 correction): integer estimate + two Newton steps
    float x = __uint_as_float(0x7EF311C7u - __float_as_uint(r));
    x = x * fmaf(-r, x, 2.0f);
    x = x * fmaf(-r, x, 2.0f);
    return 0.5f * x;
}
constexpr float kPio2Hi = 1.57079637050628662109375f, kPio2Lo = -4.37113900018624283e-8f;   // pi/2 = hi + lo
// ACOS = false: asin(x); true: acos(x).  x in [0, 1] (a NaN stays a NaN).
template <bool ACOS>
__device__ __forceinline__ float spec_asin_acos(float x) {
    const bool small = x < 0.5f;
    const float z = small ? x * x : (1.0f - x) * 0.5f;
    const float r = small ? x : sqrtf(z);
    const float e = small ? 0.0f : fmaf(-r, r, z);                 // z - r^2, exact
    const float rl = e * spec_half_rcp(fmaxf(r, 1e-20f));           // sqrt(z) = r + rl
    const float m = r * z, pz = spec_asin_poly(z);
    const float b = fmaf(m, pz, rl);                                // asin(r + rl) = r + b
    // the branch that subtracts from pi/2 (asin: x >= 0.5 with k = 2; acos: x < 0.5 with k = 1)
    const float k = ACOS ? 1.0f : 2.0f;
    const float t = fmaf(-k, r, kPio2Hi);
    const float err = fmaf(-k, r, kPio2Hi - t);                    // (pi/2_hi - k r) - t, exact (Fast2Sum: pi/2_hi >= k r)
    const float far = t + fmaf(-k, b, err + kPio2Lo);
    const float near = ACOS ? 2.0f * (r + b) : fmaf(m, pz, r);      // (asin, x < 0.5: rl = 0, one rounding)
    return (small != ACOS) ? near : far;
}
// utils/ops.py:32-44 angle2xyz(theta, phi): theta in [0, pi/2], phi in [0, 2 pi]
__device__ __forceinline__ f3 angle2xyz(float theta, float phi) {
    float st, ct, sp, cp;
    spec_sincos<true>(theta, st, ct);
    spec_sincos(phi, sp, cp);
    return t_normalize(mk3(st * cp, st * sp, ct));
}

// model/brdf.py:20-34 diffuse_sampler: theta = asin(sqrt(u0)), phi = 2 pi u1
__device__ __forceinline__ f3 diffuse_sampler(float u0, float u1, f3 n, f3 t, f3 b) {
    const float theta = spec_asin_acos<false>(sqrtf(u0));
    f3 l = angle2xyz(theta, kTwoPi * u1);
    return to_world(l, t, b, n);
}

// model/brdf.py:36-59 specular_sampler: theta = acos(sqrt(c2)), c2 = (1 - u0) / (u0 (alpha^2 - 1) + 1)
__device__ __forceinline__ f3 specular_sampler(float u0, float u1, float rough, f3 wo, f3 n, f3 t, f3 b) {
    float alpha = rough * rough;
    float c2 = (1.f - u0) / (u0 * (alpha * alpha - 1.f) + 1.f);
    const float theta = spec_asin_acos<true>(sqrtf(c2));
    f3 l = angle2xyz(theta, kTwoPi * u1);
    f3 wh = to_world(l, t, b, n);
    float s = 2.f * t_dot(wo, wh);
    return t_normalize(mk3(s * wh.x - wo.x, s * wh.y - wo.y, s * wh.z - wo.z));
}

// utils/ops.py:77-82 D_GGX, :46-63 G1_GGX_Schlick/G_Smith, :70-73 fresnelSchlick_sep
__device__ __forceinline__ float D_GGX(float cos_h, float eta) {
    float alpha = eta * eta, alpha2 = alpha * alpha;
    float denom = cos_h * cos_h * (alpha2 - 1.0f) + 1.0f;
    denom = kPi * denom * denom;
    return alpha2 / denom;
}
__device__ __forceinline__ float G1_GGX_Schlick(float NoV, float eta) {
    float k = eta + 1.f;
    k = k * k / 8.f;
    return 1.f / (NoV * (1.f - k) + k);
}
struct SpecW { float pdf, g0, g1; };
// model/brdf.py:112-136 BaseBRDF.sample_specular, the part after the direction is known
__device__ __forceinline__ SpecW specular_weights(f3 wi, f3 wo, f3 n, float rough, bool want_pdf) {
    f3 h = t_normalize(mk3(wi.x + wo.x, wi.y + wo.y, wi.z + wo.z));
    float NoL = relu(t_dot(wi, n)), NoV = relu(t_dot(wo, n));
    float VoH = relu(t_dot(wo, h)), NoH = relu(t_dot(n, h));
    SpecW r;
    r.pdf = 0.f;
    if (want_pdf) r.pdf = D_GGX(NoH, rough) / (4.f * fmaxf(VoH, 1e-4f)) * NoH;
    float G = G1_GGX_Schlick(NoL, rough) * G1_GGX_Schlick(NoV, rough);
    float x1 = 1.f - VoH, x2 = x1 * x1;
    float x = x2 * x2 * x1;  // (1-VoH).pow(5)
    float fac = G * VoH * NoL / fmaxf(NoH, 1e-4f);
    r.g0 = (1.f - x) * fac;
    r.g1 = x * fac;
    return r;
}

// ---- Philox4x32-10 (perf-mode uniforms): counter=(idx_lo,idx_hi,stream,0) key=(seed_lo,seed_hi), idx = pixel * spp + sample; outputs c0, c1 are the
// sample's (u0, u1).  (c2, c3 are a second pair, but nothing in these kernels can take it for free: a lane would have to sample two rays per block, and
// both ways of arranging that -- the pairs drawn in a pass of their own, or a lane sampling rays 2q and 2q + 1 with the second pair parked in the ray's
// slot -- measured SLOWER than drawing a block per sample, -2.5 % and -4 % of the view kernel, EXPERIMENTS.md round 4.)
__device__ __forceinline__ void philox_u2(uint64_t seed, uint64_t idx, uint32_t stream, float& u0, float& u1) {
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = stream, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: integer multiplies are the slow
        // instructions of the sampling pass
        const uint64_t m0 = (uint64_t)0xD2511F53u * (uint64_t)c0, m1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        uint32_t h0 = (uint32_t)(m0 >> 32), l0 = (uint32_t)m0;
        uint32_t h1 = (uint32_t)(m1 >> 32), l1 = (uint32_t)m1;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    }
    u0 = (float)(c0 >> 8) * 5.9604644775390625e-08f;
    u1 = (float)(c1 >> 8) * 5.9604644775390625e-08f;
}

// ---- VoxelSLF (model/slf.py) and SLFEmitter tables (model/emitter.py) as laid out in HBM ----
struct SlfDev {
    const int32_t* inds;     // H^3 int32 [z][y][x], -1 empty (the reference keeps int64: 128 MiB -> 64 MiB at H=256)
    const float4* radiance;  // kv rows padded to 16 B: one dwordx4 gather per lookup
    int H;
    float vmin, den;         // float32(voxel_min), float32(voxel_max - voxel_min)
};
struct EmitDev {
    const int32_t* emit_ord;  // nf: emitter ordinal or -1   (is_emitter + emitter_idx, model/emitter.py:153-162)
    const float4* radiance;   // n_rad rows padded to 16 B, indexed by emitter ordinal (model/emitter.py:203)
    const float* area;        // k
    int64_t nf;
    float emitter_pdf;        // 1/k
};

// model/slf.py:41-54 spatial_idx: ((x-vmin)/(vmax-vmin)*H).long().clamp(0,H-1) -> inds[z,y,x]
__device__ __forceinline__ int voxel_coord(float p, const SlfDev& s) {
    float f = (p - s.vmin) / s.den * (float)s.H;
    int v = (int)f;                    // v_cvt_i32_f32: truncates, saturates, NaN -> 0
    if (!(f < 9.2e18f)) v = 0;         // torch's int64 cast of +inf/huge/NaN is INT64_MIN -> clamps to 0
    return min(max(v, 0), s.H - 1);
}
__device__ __forceinline__ int slf_index(const SlfDev& s, f3 p) {
    int cx = voxel_coord(p.x, s), cy = voxel_coord(p.y, s), cz = voxel_coord(p.z, s);
    return s.inds[((int64_t)cz * s.H + cy) * s.H + cx];
}
// model/slf.py:63-70 forward   (j: the voxel row that was read, -1 = empty space)
__device__ __forceinline__ f3 slf_forward(const SlfDev& s, f3 p, int& j) {
    j = slf_index(s, p);
    if (j < 0) return mk3(0.f, 0.f, 0.f);
    float4 r = s.radiance[j];
    return mk3(r.x, r.y, r.z);
}
__device__ __forceinline__ f3 slf_forward(const SlfDev& s, f3 p) { int j; return slf_forward(s, p, j); }
// model/emitter.py:180-221 eval_emitter, one sample.  tri = original triangle index or -1.
// src (diagnostics): which table row the radiance came from: -2 - emitter ordinal for an emitter triangle, the VoxelSLF row for the
// radiance cache, -1 for empty space / a miss / no lookup.
// (ord_known: the emitter ordinal of `tri` when the caller has it already -- the fused bake kernels read it from the hit triangle's record, where the host put it
//  beside the vertices (iris_hip.hip fused_tris): one scattered 64-B line request per sample less; e.emit_ord is NULL then)
__device__ __forceinline__ f3 eval_emitter1(const EmitDev& e, const SlfDev& s, f3 p, int64_t tri, bool has_rough,
                                            float rough, float trace_rough, float& emit_pdf, bool& valid_next, int& src, int ord_known = -1) {
    bool vis = tri != -1;
    f3 Le = mk3(0.f, 0.f, 0.f);
    emit_pdf = 0.f;
    int ord = -1;
    if (vis) ord = e.emit_ord ? e.emit_ord[tri < 0 ? tri + e.nf : tri] : ord_known;
    bool is_area = ord >= 0;
    src = -1;
    if (is_area) {
        src = -2 - ord;
        float4 r = e.radiance[ord];
        Le = mk3(r.x, r.y, r.z);
        emit_pdf = e.emitter_pdf / fmaxf(e.area[ord], 1e-12f);
    }
    valid_next = (!is_area) && vis;
    if (has_rough && (!is_area) && vis && rough > trace_rough) {
        f3 d = slf_forward(s, p, src);
        Le = mk3(Le.x + d.x, Le.y + d.y, Le.z + d.z);
        if ((d.x + d.y) + d.z > 0.f) valid_next = false;
    }
    return Le;
}
__device__ __forceinline__ f3 eval_emitter1(const EmitDev& e, const SlfDev& s, f3 p, int64_t tri, bool has_rough,
                                            float rough, float trace_rough, float& emit_pdf, bool& valid_next) {
    int src;
    return eval_emitter1(e, s, p, tri, has_rough, rough, trace_rough, emit_pdf, valid_next, src);
}

}  // namespace iris
