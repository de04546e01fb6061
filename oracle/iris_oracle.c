/*
 * iris_oracle.c -- CPU ORACLE for the bake_shading hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product (iris_amd/,
 * libiris_hip.so) never links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference's algorithm (IEEE f32 arithmetic, libm
 * transcendentals), every function citing the reference file:line it follows.
 *
 * Pinning status
 *   - a1 ray generation, a3/a4 BRDF samplers, a5 VoxelSLF / SLFEmitter.eval_emitter, a6 MC
 *     reduction, a10 lerp_specular: PINNED against golden vectors produced by importing the
 *     reference's own Python (tools/make_goldens.py -> tests/golden/ npz files).
 *   - a2 ray/mesh intersection: PARITY UNPINNED.  The reference delegates it to Mitsuba 3.5.0
 *     (cuda_ad_rgb / OptiX; environment.yml:14-15), whose source is not under /root/reference
 *     and which the reference has no test vectors for.  OptiX's triangle test is closed; what is known
 *     about it is that it is watertight, so the hit decision restated here is the published watertight test
 *     (Woop, Benthin, Wald, "Watertight Ray/Triangle Intersection", JCGT 2(1), 2013, section 3; tri_test
 *     below), with Mitsuba 3's Mesh::compute_surface_interaction conventions for what is returned
 *     (si.p = barycentric interpolation, si.n = unit geometric normal, si.uv = (b1,b2) when the
 *     mesh carries no texcoords, t = +inf on a miss), anchored on the reference's call site
 *     utils/path_tracing.py:17-48.
 *
 * Arithmetic contract for the intersection (so that a different implementation can be compared
 * bit for bit): all operations are IEEE-754 binary32, no contraction other than the explicit
 * fmaf() calls written below (compile with -ffp-contract=off):
 *     cross(a,b) = ( fma(a.y,b.z,-(a.z*b.y)), fma(a.z,b.x,-(a.x*b.z)), fma(a.x,b.y,-(a.y*b.x)) )      (geometric normal)
 *     dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *     ray-space row:  fma(m.x,p.x, fma(m.y,p.y, fma(m.z,p.z, c)));   edge functions: plain products, plain difference (tri_test)
 * Closest hit = lexicographic minimum of (t, triangle index) over all triangles that pass the test.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_ld(const float *p) { return v3_make(p[0], p[1], p[2]); }
static inline void v3_st(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* ---- arithmetic contract (see header) ---- */
static inline v3 x_cross(v3 a, v3 b) {
    return v3_make(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float x_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }

/* torch-style helpers: sum over the last dim in index order, no fma */
static inline float t_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* NF.normalize(v,dim=-1): v / max(||v||_2, 1e-12) */
static inline v3 t_normalize(v3 a) {
    float n = sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
    if (n < 1e-12f) n = 1e-12f;
    return v3_make(a.x / n, a.y / n, a.z / n);
}
/* torch.cross(a,b,dim=-1) */
static inline v3 t_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float relu(float x) { return x > 0.f ? x : 0.f; }

static const float PI_F = 3.14159265358979323846f;      /* float32(math.pi)   */
static const float TWO_PI_F = 6.28318530717958647692f;  /* float32(2*math.pi) */

/* ============================================================================================
 * a1  ray generation
 * ========================================================================================== */

/* utils/dataset/real_ldr.py:49-61 get_direction + :63-83 to_world (also the ScanNet++ loader,
 * utils/dataset/scannetpp/dataset.py:202-215).  K row-major 3x3, c2w row-major 3x4.
 * rays_o/rays_d: (H*W,3) row-major, x fastest.  ray_diff!=0: un-normalised d plus dxdu,dydv. */
ORC_API void orc_raygen_real(const float *K, const float *c2w, int H, int W, int ray_diff,
                             float *rays_o, float *rays_d, float *dxdu, float *dydv) {
    const float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int64_t i = (int64_t)y * W + x;
            /* torch.linspace(0.5,N-0.5,N) has step exactly 1 -> pixel centre x+0.5 */
            float dc[3] = {((float)x + 0.5f - cx) / fx, ((float)y + 0.5f - cy) / fy, 1.0f};
            v3 d;
            /* rays_d @ c2w[:3,:3].T  -> d[i] = sum_j dc[j]*R[i][j] */
            d.x = (dc[0] * c2w[0] + dc[1] * c2w[1]) + dc[2] * c2w[2];
            d.y = (dc[0] * c2w[4] + dc[1] * c2w[5]) + dc[2] * c2w[6];
            d.z = (dc[0] * c2w[8] + dc[1] * c2w[9]) + dc[2] * c2w[10];
            rays_o[i * 3 + 0] = c2w[3]; rays_o[i * 3 + 1] = c2w[7]; rays_o[i * 3 + 2] = c2w[11];
            if (ray_diff) {
                v3_st(rays_d + i * 3, d);
                float ifx = 1.0f / fx, ify = 1.0f / fy;
                dxdu[i * 3 + 0] = ifx * c2w[0]; dxdu[i * 3 + 1] = ifx * c2w[4]; dxdu[i * 3 + 2] = ifx * c2w[8];
                dydv[i * 3 + 0] = ify * c2w[1]; dydv[i * 3 + 1] = ify * c2w[5]; dydv[i * 3 + 2] = ify * c2w[9];
            } else {
                v3_st(rays_d + i * 3, t_normalize(d));
            }
        }
}

/* utils/dataset/synthetic_ldr.py:21-34 get_ray_directions + :36-57 get_rays.
 * focal_diff!=0 corresponds to get_rays(..., focal=focal): un-normalised d + differentials. */
ORC_API void orc_raygen_synthetic(float focal, const float *c2w, int H, int W, int ray_diff,
                                  float *rays_o, float *rays_d, float *dxdu, float *dydv) {
    /* W/2 and H/2 are python float divisions */
    const float hw = (float)((double)W / 2.0), hh = (float)((double)H / 2.0);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int64_t i = (int64_t)y * W + x;
            float dc[3] = {-(((float)x + 0.5f) - hw) / focal, -(((float)y + 0.5f) - hh) / focal, 1.0f};
            v3 d;
            d.x = (dc[0] * c2w[0] + dc[1] * c2w[1]) + dc[2] * c2w[2];
            d.y = (dc[0] * c2w[4] + dc[1] * c2w[5]) + dc[2] * c2w[6];
            d.z = (dc[0] * c2w[8] + dc[1] * c2w[9]) + dc[2] * c2w[10];
            rays_o[i * 3 + 0] = c2w[3]; rays_o[i * 3 + 1] = c2w[7]; rays_o[i * 3 + 2] = c2w[11];
            if (ray_diff) {
                v3_st(rays_d + i * 3, d);
                float inv = 1.0f / focal;
                dxdu[i * 3 + 0] = inv * c2w[0]; dxdu[i * 3 + 1] = inv * c2w[4]; dxdu[i * 3 + 2] = inv * c2w[8];
                dydv[i * 3 + 0] = inv * c2w[1]; dydv[i * 3 + 1] = inv * c2w[5]; dydv[i * 3 + 2] = inv * c2w[9];
            } else {
                float n = sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z); /* rays_d / torch.norm(rays_d) */
                v3_st(rays_d + i * 3, v3_make(d.x / n, d.y / n, d.z / n));
            }
        }
}

/* ============================================================================================
 * a3/a4  shading math (utils/ops.py) and BRDF samplers (model/brdf.py)
 * ========================================================================================== */

/* utils/ops.py:12-30 get_normal_space -> columns (tangent, bitangent, normal) */
static inline void normal_space(v3 n, v3 *t, v3 *b) {
    /* (v1*normal).sum(-1).abs() <= 1e-1, v1=(1,0,0); 1e-1 is compared in float32 */
    if (fabsf(n.x) <= 0.1f) *t = t_normalize(t_cross(v3_make(1.f, 0.f, 0.f), n));
    else                    *t = t_normalize(t_cross(v3_make(0.f, 1.f, 0.f), n));
    *b = t_cross(n, *t);
}
ORC_API void orc_get_normal_space(const float *normal, int64_t B, float *out /* B*3*3, [i][j], j=t,b,n */) {
    for (int64_t i = 0; i < B; ++i) {
        v3 n = v3_ld(normal + i * 3), t, b;
        normal_space(n, &t, &b);
        float *o = out + i * 9;
        o[0] = t.x; o[1] = b.x; o[2] = n.x;
        o[3] = t.y; o[4] = b.y; o[5] = n.y;
        o[6] = t.z; o[7] = b.z; o[8] = n.z;
    }
}
/* utils/ops.py:32-44 angle2xyz */
static inline v3 angle2xyz(float theta, float phi) {
    float st = sinf(theta);
    return t_normalize(v3_make(st * cosf(phi), st * sinf(phi), cosf(theta)));
}
/* (wi[:,None] @ Nmat.permute(0,2,1)).squeeze(1): out = l.x*t + l.y*b + l.z*n */
static inline v3 to_world(v3 l, v3 t, v3 b, v3 n) {
    return v3_make((l.x * t.x + l.y * b.x) + l.z * n.x, (l.x * t.y + l.y * b.y) + l.z * n.y,
                   (l.x * t.z + l.y * b.z) + l.z * n.z);
}
/* ---- "device arithmetic" mode -------------------------------------------------------------------------------
 * mode 0 (default): the literal restatement above/below (libm asinf/acosf/sinf/cosf/powf) -- this is what is pinned
 *                   against the reference's golden vectors.
 * mode 1          : the same algorithm with the transcendental round trips replaced by the exactly specified IEEE
 *                   operation sequences the HIP kernels use (iris_amd/csrc/iris_device.h): a specified asin / acos on
 *                   [0, 1] (spec_asin_acos: explicit fmaf, IEEE sqrtf, an integer-seeded Newton reciprocal; within 1 ulp
 *                   of the correctly rounded value and equal to it for 98.5 % of the 2^24 Philox outputs), a double-precision
 *                   (explicit fma, practically correctly rounded)
 *                   polynomial sincos, x^5 by multiplication, and the kernels' fixed reduction order for the mean over
 *                   spp.  Mode 1 exists so that GPU results can be compared BIT FOR BIT at any size (the SLF / emitter
 *                   lookups are discontinuous, so 1-ulp differences in a direction flip rare samples by O(1)); it is
 *                   itself checked against mode 0 / the goldens within float tolerance by the CPU tests. */
static int g_mode = 0;
ORC_API void orc_set_mode(int m) { g_mode = m; }
ORC_API int orc_get_mode(void) { return g_mode; }
/* Attribution studies (tools/attribute_flips.py): in mode 1, undo single substitutions -- bit 0: libm asinf / acosf for the polar angle, bit 1: libm
 * sinf / cosf of the polar angle, bit 2: libm sinf / cosf of the azimuth, bit 3: powf for x^5 -- to see which one a disagreement with the reference comes from. */
static int g_undo = 0;
ORC_API void orc_set_undo(int m) { g_undo = m; }

/* spec_sincos of iris_device.h, operation for operation: quadrant reduction and both kernels in double (explicit fma), one rounding to f32 */
static inline void spec_sincos(float x, float *s, float *c) {
    const int j = (int)(x * 0.636619772367581343f + 0.5f);
    const double z = fma(-(double)j, 1.57079632679489661923, (double)x);
    const double zz = z * z;
    double p = 2.7249902524065394e-06;
    p = fma(p, zz, -0.0001984008661425884); p = fma(p, zz, 0.00833333187464819); p = fma(p, zz, -0.16666666663855825);
    const float ps = (float)fma(z * zz, p, z);
    double q = -2.723710465738025e-07;
    q = fma(q, zz, 2.4799861845569796e-05); q = fma(q, zz, -0.0013888885090442048); q = fma(q, zz, 0.04166666663738883); q = fma(q, zz, -0.4999999999996389);
    const float pc = (float)fma(zz, q, 1.0);
    const int k = j & 3;
    *s = (k == 0) ? ps : (k == 1) ? pc : (k == 2) ? -ps : -pc;
    *c = (k == 0) ? pc : (k == 1) ? -ps : (k == 2) ? -pc : ps;
}
/* spec_asin_acos of iris_device.h, operation for operation (x in [0, 1]) */
static inline float as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float spec_asin_poly(float z) {
    float p = 0.033805747f;
    p = fmaf(p, z, 0.01707786f); p = fmaf(p, z, 0.031116156f); p = fmaf(p, z, 0.04459803f); p = fmaf(p, z, 0.07500099f);
    return fmaf(p, z, 0.16666666f);
}
static inline float spec_half_rcp(float r) {
    float x = as_f32(0x7EF311C7u - as_u32(r));
    x = x * fmaf(-r, x, 2.0f);
    x = x * fmaf(-r, x, 2.0f);
    return 0.5f * x;
}
static const float PIO2_HI = 1.57079637050628662109375f, PIO2_LO = -4.37113900018624283e-8f;
static inline float spec_asin_acos(float x, int acos_) {
    const int small = x < 0.5f;
    const float z = small ? x * x : (1.0f - x) * 0.5f;
    const float r = small ? x : sqrtf(z);
    const float e = small ? 0.0f : fmaf(-r, r, z);
    const float rl = e * spec_half_rcp(fmaxf(r, 1e-20f));
    const float m = r * z, pz = spec_asin_poly(z);
    const float b = fmaf(m, pz, rl);
    const float k = acos_ ? 1.0f : 2.0f;
    const float t = fmaf(-k, r, PIO2_HI);
    const float err = fmaf(-k, r, PIO2_HI - t);
    const float far_ = t + fmaf(-k, b, err + PIO2_LO);
    const float near_ = acos_ ? 2.0f * (r + b) : fmaf(m, pz, r);
    return (small != (acos_ != 0)) ? near_ : far_;
}
ORC_API void orc_spec_asin_acos(const float *x, int64_t n, int acos_, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = spec_asin_acos(x[i], acos_);
}
/* angle2xyz through the specified sincos (theta in [0, pi/2]: the first-quadrant selects of the kernel give the same values) */
static inline v3 angle2xyz_spec(float theta, float phi) {
    float st, ct, sp, cp;
    if (g_undo & 2) { st = sinf(theta); ct = cosf(theta); } else spec_sincos(theta, &st, &ct);
    if (g_undo & 4) { sp = sinf(phi); cp = cosf(phi); } else spec_sincos(phi, &sp, &cp);
    return t_normalize(v3_make(st * cp, st * sp, ct));
}

/* utils/ops.py:85-96 double_sided(V,N) (in place on N) */
ORC_API void orc_double_sided(const float *V, float *N, int64_t B) {
    for (int64_t i = 0; i < B; ++i) {
        v3 v = v3_ld(V + i * 3), n = v3_ld(N + i * 3);
        if (t_dot(n, v) < 0.f) { N[i * 3] = -n.x; N[i * 3 + 1] = -n.y; N[i * 3 + 2] = -n.z; }
    }
}

/* model/brdf.py:20-34 diffuse_sampler */
static inline v3 diffuse_sampler(float u0, float u1, v3 n) {
    float phi = TWO_PI_F * u1;
    v3 l, t, b;
    if (g_mode == 1) {
        l = angle2xyz_spec((g_undo & 1) ? asinf(sqrtf(u0)) : spec_asin_acos(sqrtf(u0), 0), phi);
    } else {
        float theta = asinf(sqrtf(u0));
        l = angle2xyz(theta, phi);
    }
    normal_space(n, &t, &b);
    return to_world(l, t, b, n);
}
/* model/brdf.py:78-88 BaseBRDF.sample_diffuse */
ORC_API void orc_sample_diffuse(const float *u2, const float *normal, int64_t B, float *wi, float *pdf, float *w) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < B; ++i) {
        v3 n = v3_ld(normal + i * 3);
        v3 d = diffuse_sampler(u2[i * 2], u2[i * 2 + 1], n);
        v3_st(wi + i * 3, d);
        if (pdf) pdf[i] = relu(t_dot(n, d)) / PI_F;
        if (w) { w[i * 3] = 1.f; w[i * 3 + 1] = 1.f; w[i * 3 + 2] = 1.f; }
    }
}

/* model/brdf.py:36-59 specular_sampler */
static inline v3 specular_sampler(float u0, float u1, float rough, v3 wo, v3 n) {
    float alpha = rough * rough;
    float c2 = (1.f - u0) / (u0 * (alpha * alpha - 1.f) + 1.f);
    float phi = TWO_PI_F * u1;
    v3 l, t, b;
    if (g_mode == 1) {
        l = angle2xyz_spec((g_undo & 1) ? acosf(sqrtf(c2)) : spec_asin_acos(sqrtf(c2), 1), phi);
    } else {
        float theta = acosf(sqrtf(c2));
        l = angle2xyz(theta, phi);
    }
    normal_space(n, &t, &b);
    v3 wh = to_world(l, t, b, n);
    float s = 2.f * t_dot(wo, wh);
    v3 wi = v3_make(s * wh.x - wo.x, s * wh.y - wo.y, s * wh.z - wo.z);
    return t_normalize(wi);
}
/* utils/ops.py:77-82 D_GGX */
static inline float D_GGX(float cos_h, float eta) {
    float alpha = eta * eta, alpha2 = alpha * alpha;
    float denom = cos_h * cos_h * (alpha2 - 1.0f) + 1.0f;
    denom = PI_F * denom * denom;
    return alpha2 / denom;
}
/* utils/ops.py:46-63 G1_GGX_Schlick / G_Smith */
static inline float G1_GGX_Schlick(float NoV, float eta) {
    float k = eta + 1.f;
    k = k * k / 8.f;
    return 1.f / (NoV * (1.f - k) + k);
}
static inline float pow5(float x) {
    if (g_mode == 1 && !(g_undo & 8)) { float x2 = x * x; return x2 * x2 * x; }
    return powf(x, 5.f);
}

typedef struct { v3 wi; float pdf, g0, g1; } spec_sample;
/* model/brdf.py:112-136 BaseBRDF.sample_specular */
static inline spec_sample sample_specular1(float u0, float u1, v3 wo, v3 n, float rough) {
    spec_sample r;
    r.wi = specular_sampler(u0, u1, rough, wo, n);
    v3 h = t_normalize(v3_make(r.wi.x + wo.x, r.wi.y + wo.y, r.wi.z + wo.z));
    float NoL = relu(t_dot(r.wi, n)), NoV = relu(t_dot(wo, n));
    float VoH = relu(t_dot(wo, h)), NoH = relu(t_dot(n, h));
    float D = D_GGX(NoH, rough);
    float vc = VoH < 1e-4f ? 1e-4f : VoH;
    r.pdf = D / (4.f * vc) * NoH;
    float G = G1_GGX_Schlick(NoL, rough) * G1_GGX_Schlick(NoV, rough);
    float x = pow5(1.f - VoH);
    float F0 = 1.f - x, F1 = x;
    float nc = NoH < 1e-4f ? 1e-4f : NoH;
    float fac = G * VoH * NoL / nc;
    r.g0 = F0 * fac; r.g1 = F1 * fac;
    return r;
}
ORC_API void orc_sample_specular(const float *u2, const float *wo, const float *normal, float rough, int64_t B,
                                 float *wi, float *pdf, float *g0, float *g1) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < B; ++i) {
        spec_sample r = sample_specular1(u2[i * 2], u2[i * 2 + 1], v3_ld(wo + i * 3), v3_ld(normal + i * 3), rough);
        v3_st(wi + i * 3, r.wi);
        if (pdf) pdf[i] = r.pdf;
        if (g0) g0[i] = r.g0;
        if (g1) g1[i] = r.g1;
    }
}

/* utils/ops.py:99-118 lerp_specular: specular (B,R,3), roughness (B) */
ORC_API void orc_lerp_specular(const float *spec, const float *rough, int64_t B, int R, float *out) {
    for (int64_t i = 0; i < B; ++i) {
        /* (roughness-r_min)/(r_max-r_min)*(r_num-1); r_max-r_min is a python double 0.98 cast to f32 */
        float r = (rough[i] - 0.02f) / (float)(1.0 - 0.02) * (float)(R - 1);
        int64_t r1 = (int64_t)ceilf(r), r0 = (int64_t)floorf(r);
        float w = r - (float)r0;
        for (int c = 0; c < 3; ++c) {
            float s0 = spec[(i * R + r0) * 3 + c], s1 = spec[(i * R + r1) * 3 + c];
            out[i * 3 + c] = s0 * (1.f - w) + s1 * w;
        }
    }
}

/* --------------------------------------------------------------------------------------------
 * 8(f)-3  packed shading cache + the BRDF trainer's shading combine
 * Packing (utils/dataset/scannetpp/dataset.py:359-377): per pixel diffuse(3) | spec0 level 0..R-1 (3 each) | spec1 level 0..R-1.
 * The oracle keeps the reference's (3+6R)-float row; the HIP library keeps its own padded / level-interleaved row and must
 * return the same values.  Slice (:409-414) = row gather by pixel index.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_cache_pack(const float *diffuse, const float *const *spec0, const float *const *spec1, int64_t n, int R, float *rows) {
    const int S = 3 + 6 * R;
    for (int64_t i = 0; i < n; ++i) {
        float *q = rows + i * S;
        for (int c = 0; c < 3; ++c) q[c] = diffuse[i * 3 + c];
        for (int j = 0; j < R; ++j)
            for (int c = 0; c < 3; ++c) { q[3 + j * 3 + c] = spec0[j][i * 3 + c]; q[3 + 3 * R + j * 3 + c] = spec1[j][i * 3 + c]; }
    }
}
ORC_API void orc_cache_gather(const float *rows, const int64_t *idx, int64_t B, int R, float *out) {
    const int S = 3 + 6 * R;
    for (int64_t i = 0; i < B; ++i) memcpy(out + i * S, rows + (idx ? idx[i] : i) * S, (size_t)S * sizeof(float));
}

typedef struct { int64_t r0, r1; float w; } lerp_pos;
static lerp_pos lerp_position(float rough, int R) { /* utils/ops.py:108-115 */
    lerp_pos p;
    float r = (rough - 0.02f) / (float)(1.0 - 0.02) * (float)(R - 1);
    p.r1 = (int64_t)ceilf(r); p.r0 = (int64_t)floorf(r);
    p.w = r - (float)p.r0;
    /* out-of-range roughness: clamp (the reference would index out of range) */
    if (p.r0 < 0) p.r0 = 0;
    if (p.r1 < 0) p.r1 = 0;
    if (p.r0 > R - 1) p.r0 = R - 1;
    if (p.r1 > R - 1) p.r1 = R - 1;
    return p;
}

/* train_brdf_crf.py:195-203: kd = albedo*(1-metallic); ks = 0.04*(1-metallic) + albedo*metallic;
 * L = kd*diffuse + ks*lerp_specular(specular0, roughness) + lerp_specular(specular1, roughness)     (rows in the (3+6R) layout) */
ORC_API void orc_shade_cached_fwd(const float *rows, const int64_t *idx, const float *albedo, const float *metallic, const float *roughness,
                                  int64_t B, int R, float *L) {
    const int S = 3 + 6 * R;
    for (int64_t i = 0; i < B; ++i) {
        const float *q = rows + (idx ? idx[i] : i) * S;
        lerp_pos p = lerp_position(roughness[i], R);
        float m = metallic[i], m1 = 1.f - m;
        for (int c = 0; c < 3; ++c) {
            float a = albedo[i * 3 + c];
            float kd = a * m1, ks = 0.04f * m1 + a * m;
            float S0 = q[3 + p.r0 * 3 + c] * (1.f - p.w) + q[3 + p.r1 * 3 + c] * p.w;
            float S1 = q[3 + 3 * R + p.r0 * 3 + c] * (1.f - p.w) + q[3 + 3 * R + p.r1 * 3 + c] * p.w;
            float Ld = kd * q[c], Ls = ks * S0 + S1;
            L[i * 3 + c] = Ld + Ls;
        }
    }
}
/* gradient of the above w.r.t. albedo (B,3), metallic (B), roughness (B) for an incoming gL (B,3); fixed summation order c = 0,1,2 */
ORC_API void orc_shade_cached_bwd(const float *rows, const int64_t *idx, const float *albedo, const float *metallic, const float *roughness,
                                  const float *gL, int64_t B, int R, float *g_albedo, float *g_metallic, float *g_roughness) {
    const int S = 3 + 6 * R;
    for (int64_t i = 0; i < B; ++i) {
        const float *q = rows + (idx ? idx[i] : i) * S;
        lerp_pos p = lerp_position(roughness[i], R);
        float m = metallic[i], m1 = 1.f - m;
        float g_m = 0.f, g_m1 = 0.f, g_w = 0.f;
        for (int c = 0; c < 3; ++c) {
            float a = albedo[i * 3 + c], g = gL[i * 3 + c];
            float s0a = q[3 + p.r0 * 3 + c], s0b = q[3 + p.r1 * 3 + c];
            float s1a = q[3 + 3 * R + p.r0 * 3 + c], s1b = q[3 + 3 * R + p.r1 * 3 + c];
            float S0 = s0a * (1.f - p.w) + s0b * p.w;
            float ks = 0.04f * m1 + a * m;
            float g_kd = g * q[c], g_ks = g * S0;
            g_albedo[i * 3 + c] = g_kd * m1 + g_ks * m;
            g_m += g_ks * a;
            g_m1 += g_kd * a + g_ks * 0.04f;
            g_w += (g * ks) * (s0b - s0a) + g * (s1b - s1a);
        }
        g_metallic[i] = g_m - g_m1;
        g_roughness[i] = g_w * (float)(R - 1) / (float)(1.0 - 0.02);
    }
}

/* --------------------------------------------------------------------------------------------
 * 8(f)-4  denoiser substitute.  NO reference counterpart can be restated: bake_shading.py:81,129,198-200 call the closed OptiX AI
 * denoiser.  This is the CPU restatement of the BUILD'S OWN filter (iris_amd/csrc/iris_denoise.h: variance-guided edge-avoiding
 * a-trous), used to check the HIP kernels tap for tap (tolerance: expf / log2f differ in the last ulp between libm and the device);
 * quality is judged separately on PSNR against a high-spp bake.  "parity unpinned" by construction.
 * ------------------------------------------------------------------------------------------ */
static float dn_lum(const float *c) { return 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2]; }
typedef struct { int H, W; float sigma_l, sigma_n, sigma_p; const float *normal, *position; const uint8_t *valid; } dn_ctx;
static int dn_valid(const dn_ctx *d, int64_t q) { return d->valid ? d->valid[q] != 0 : 1; }
static void dn_guide(const dn_ctx *d, int64_t q, float *n, float *x) {
    if (d->normal && dn_valid(d, q)) { n[0] = d->normal[q * 3]; n[1] = d->normal[q * 3 + 1]; n[2] = d->normal[q * 3 + 2]; } else { n[0] = 0.f; n[1] = 0.f; n[2] = 1.f; }
    if (d->position && dn_valid(d, q)) { x[0] = d->position[q * 3]; x[1] = d->position[q * 3 + 1]; x[2] = d->position[q * 3 + 2]; } else { x[0] = x[1] = x[2] = 0.f; }
}
static float dn_geo(const dn_ctx *d, int64_t p, int64_t q) {
    if (!dn_valid(d, q)) return 0.f;
    float np[3], xp[3], nq[3], xq[3];
    dn_guide(d, p, np, xp); dn_guide(d, q, nq, xq);
    float nn = fmaxf(0.f, np[0] * nq[0] + np[1] * nq[1] + np[2] * nq[2]);
    float wn = nn > 0.f ? exp2f(d->sigma_n * log2f(nn)) : 0.f;
    float dx = xq[0] - xp[0], dy = xq[1] - xp[1], dz = xq[2] - xp[2];
    float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    float plane = fabsf(np[0] * dx + np[1] * dy + np[2] * dz);
    return wn * expf(-plane / (d->sigma_p * dist + 1e-12f));
}
/* one map: in (H*W,3) -> out (H*W,3) */
ORC_API void orc_denoise(const float *normal, const float *position, const uint8_t *valid, int H, int W, const float *in, float *out,
                         int iterations, float sigma_l, float sigma_n, float sigma_p) {
    dn_ctx d = {H, W, sigma_l, sigma_n, sigma_p, normal, position, valid};
    const int64_t n = (int64_t)H * W;
    float *a = (float *)calloc((size_t)n * 4, sizeof(float)), *b = (float *)calloc((size_t)n * 4, sizeof(float));
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int64_t p = (int64_t)y * W + x;
            if (!dn_valid(&d, p)) continue;
            float ws = 0.f, m1 = 0.f, m2 = 0.f;
            for (int dy = -3; dy <= 3; ++dy) {
                int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -3; dx <= 3; ++dx) {
                    int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    int64_t q = (int64_t)yy * W + xx;
                    float w = (dx == 0 && dy == 0) ? 1.f : dn_geo(&d, p, q);
                    if (w == 0.f) continue;
                    float l = dn_lum(in + q * 3);
                    ws += w; m1 += w * l; m2 += w * l * l;
                }
            }
            float mean = m1 / ws;
            a[p * 4] = in[p * 3]; a[p * 4 + 1] = in[p * 3 + 1]; a[p * 4 + 2] = in[p * 3 + 2];
            a[p * 4 + 3] = fmaxf(0.f, m2 / ws - mean * mean);
        }
    const float h1[3] = {3.f / 8.f, 1.f / 4.f, 1.f / 16.f};
    const float hc = h1[0] * h1[0];
    for (int it = 0; it < iterations; ++it) {
        const int step = 1 << it;
#pragma omp parallel for schedule(dynamic, 4)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int64_t p = (int64_t)y * W + x;
                if (!dn_valid(&d, p)) { b[p * 4] = b[p * 4 + 1] = b[p * 4 + 2] = b[p * 4 + 3] = 0.f; continue; }
                float gw = 0.f, gv = 0.f;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        int xx = x + dx, yy = y + dy;
                        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                        int64_t q = (int64_t)yy * W + xx;
                        if (!dn_valid(&d, q)) continue;
                        float k = (dx == 0 ? 2.f : 1.f) * (dy == 0 ? 2.f : 1.f);
                        gw += k; gv += k * a[q * 4 + 3];
                    }
                float lp = dn_lum(a + p * 4);
                float inv_sl = 1.f / (sigma_l * sqrtf(fmaxf(0.f, gv / gw)) + 1e-6f);
                float sr = a[p * 4], sg = a[p * 4 + 1], sb = a[p * 4 + 2], sv = a[p * 4 + 3], sw = 1.f;
                for (int j = -2; j <= 2; ++j) {
                    int yy = y + j * step;
                    if (yy < 0 || yy >= H) continue;
                    for (int i = -2; i <= 2; ++i) {
                        int xx = x + i * step;
                        if (xx < 0 || xx >= W || (i == 0 && j == 0)) continue;
                        int64_t q = (int64_t)yy * W + xx;
                        float wg = dn_geo(&d, p, q);
                        if (wg == 0.f) continue;
                        float h = h1[i < 0 ? -i : i] * h1[j < 0 ? -j : j] / hc * wg;
                        float w = h * expf(-fabsf(lp - dn_lum(a + q * 4)) * inv_sl);
                        sr += w * a[q * 4]; sg += w * a[q * 4 + 1]; sb += w * a[q * 4 + 2]; sv += w * w * a[q * 4 + 3]; sw += w;
                    }
                }
                float inv = 1.f / sw;
                b[p * 4] = sr * inv; b[p * 4 + 1] = sg * inv; b[p * 4 + 2] = sb * inv; b[p * 4 + 3] = sv * inv * inv;
            }
        float *t = a; a = b; b = t;
    }
    for (int64_t p = 0; p < n; ++p) { out[p * 3] = a[p * 4]; out[p * 3 + 1] = a[p * 4 + 1]; out[p * 3 + 2] = a[p * 4 + 2]; }
    free(a); free(b);
}

/* ============================================================================================
 * a5  VoxelSLF (model/slf.py) and SLFEmitter.eval_emitter (model/emitter.py)
 * ========================================================================================== */
typedef struct {
    int H;
    const int64_t *inds;   /* H^3, [z][y][x], -1 = empty */
    const float *radiance; /* kv*3 */
    int64_t kv;
    float vmin, den;       /* float32(voxel_min), float32(voxel_max - voxel_min) (python double subtraction) */
} orc_slf;

ORC_API orc_slf *orc_slf_create(const int64_t *inds, int H, const float *radiance, int64_t kv, double vmin, double vmax) {
    orc_slf *s = (orc_slf *)calloc(1, sizeof(orc_slf));
    s->H = H; s->inds = inds; s->radiance = radiance; s->kv = kv;
    s->vmin = (float)vmin; s->den = (float)(vmax - vmin);
    return s;
}
ORC_API void orc_slf_destroy(orc_slf *s) { free(s); }

/* model/slf.py:41-54 spatial_idx */
static inline int64_t slf_spatial_idx(const orc_slf *s, v3 p) {
    float q[3] = {(p.x - s->vmin) / s->den, (p.y - s->vmin) / s->den, (p.z - s->vmin) / s->den};
    int64_t c[3];
    for (int k = 0; k < 3; ++k) {
        float f = q[k] * (float)s->H;
        int64_t v;
        /* .long(): truncation toward zero; out-of-range/NaN is UB in C, saturate explicitly
           (torch on x86 yields INT64_MIN, which the clamp maps to 0) */
        if (!(f > -9.2e18f)) v = INT64_MIN; else if (f >= 9.2e18f) v = INT64_MIN; else v = (int64_t)f;
        if (v < 0) v = 0;
        if (v > s->H - 1) v = s->H - 1;
        c[k] = v;
    }
    return s->inds[(c[2] * s->H + c[1]) * s->H + c[0]];
}
ORC_API void orc_slf_spatial_idx(const orc_slf *s, const float *x, int64_t B, int64_t *idx) {
    for (int64_t i = 0; i < B; ++i) idx[i] = slf_spatial_idx(s, v3_ld(x + i * 3));
}
/* model/slf.py:63-70 forward: radiance[idx], zero where idx==-1 */
static inline v3 slf_forward(const orc_slf *s, v3 p) {
    int64_t j = slf_spatial_idx(s, p);
    if (j < 0) return v3_make(0.f, 0.f, 0.f);
    return v3_ld(s->radiance + j * 3);
}
ORC_API void orc_slf_forward(const orc_slf *s, const float *x, int64_t B, float *rgb) {
    for (int64_t i = 0; i < B; ++i) v3_st(rgb + i * 3, slf_forward(s, v3_ld(x + i * 3)));
}

typedef struct {
    int64_t nf, k;
    const uint8_t *is_emitter; /* nf */
    int64_t *emitter_idx;      /* nf, -1 = not an emitter (model/emitter.py:160-162) */
    const float *radiance;     /* rows indexed by emitter ordinal (model/emitter.py:203) */
    const float *area;         /* k */
    float emitter_pdf;         /* NF.normalize(ones(k),p=1) = 1/k (model/emitter.py:169) */
    const float *verts;        /* (k,3,3) emitter_vertices, optional (sample_emitter) */
    const float *cdf;          /* (k) emitter_cdf as torch computed it */
    int64_t *ord2tri;          /* (k) triangle_idx buffer (model/emitter.py:165-166) */
} orc_emitter;

ORC_API orc_emitter *orc_emitter_create(const uint8_t *is_emitter, int64_t nf, const float *radiance, const float *area, int64_t k) {
    orc_emitter *e = (orc_emitter *)calloc(1, sizeof(orc_emitter));
    e->nf = nf; e->k = k; e->is_emitter = is_emitter; e->radiance = radiance; e->area = area;
    e->emitter_idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nf > 0 ? nf : 1));
    int64_t c = 0;
    for (int64_t i = 0; i < nf; ++i) e->emitter_idx[i] = is_emitter[i] ? c++ : -1;
    float s = (float)k; if (s < 1e-12f) s = 1e-12f;
    e->emitter_pdf = 1.0f / s;
    e->ord2tri = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
    for (int64_t i = 0; i < nf; ++i) if (e->emitter_idx[i] >= 0 && e->emitter_idx[i] < k) e->ord2tri[e->emitter_idx[i]] = i;
    return e;
}
ORC_API void orc_emitter_set_sampling(orc_emitter *e, const float *verts, const float *cdf) { e->verts = verts; e->cdf = cdf; }
ORC_API void orc_emitter_destroy(orc_emitter *e) { if (e) { free(e->emitter_idx); free(e->ord2tri); free(e); } }

/* model/emitter.py:180-221 eval_emitter for one sample.  rough<0 encodes roughness=None. */
static inline v3 eval_emitter1(const orc_emitter *e, const orc_slf *s, v3 p, int64_t tri, int has_rough, float rough,
                               float trace_rough, float *emit_pdf, uint8_t *valid_next) {
    int vis = tri != -1;
    v3 Le = v3_make(0.f, 0.f, 0.f);
    float pdf = 0.f;
    int64_t ti = tri < 0 ? tri + e->nf : tri; /* python negative index wraps */
    int is_area = vis && e->is_emitter[ti];
    if (is_area) {
        int64_t ei = e->emitter_idx[ti];
        float a = e->area[ei]; if (a < 1e-12f) a = 1e-12f;
        pdf = e->emitter_pdf / a;
        Le = v3_ld(e->radiance + ei * 3);
    }
    int vn = (!is_area) && vis;
    if (has_rough) {
        int is_diffuse = (!is_area) && vis && (rough > trace_rough);
        if (is_diffuse) {
            v3 d = slf_forward(s, p);
            Le = v3_make(Le.x + d.x, Le.y + d.y, Le.z + d.z);
            if ((d.x + d.y) + d.z > 0.f) vn = 0;
        }
    }
    if (emit_pdf) *emit_pdf = pdf;
    if (valid_next) *valid_next = (uint8_t)vn;
    return Le;
}
ORC_API void orc_eval_emitter(const orc_emitter *e, const orc_slf *s, const float *pos, const int64_t *tri,
                              const float *rough /* nullable */, float trace_rough, int64_t B,
                              float *Le, float *emit_pdf, uint8_t *valid_next) {
    for (int64_t i = 0; i < B; ++i) {
        float pdf; uint8_t vn;
        v3 l = eval_emitter1(e, s, v3_ld(pos + i * 3), tri[i], rough != NULL, rough ? rough[i] : 0.f, trace_rough, &pdf, &vn);
        v3_st(Le + i * 3, l);
        if (emit_pdf) emit_pdf[i] = pdf;
        if (valid_next) valid_next[i] = vn;
    }
}

/* ============================================================================================
 * a2  ray / triangle-mesh closest hit  (utils/path_tracing.py:17-48; Mitsuba semantics restated)
 * ========================================================================================== */
typedef struct {
    float lo[3], hi[3];
    int32_t left;   /* internal: index of left child (right = left+1); leaf: -1 */
    int32_t start, count;
} orc_node;

typedef struct {
    int64_t nv, nf;
    const float *verts;
    const int32_t *faces;
    orc_node *nodes; int32_t n_nodes;
    int32_t *order; /* leaf triangle order */
} orc_scene;

typedef struct { float t, u, v; int64_t tri; } orc_hit;

/* Watertight ray / triangle test (Woop, Benthin, Wald 2013, "Watertight Ray/Triangle Intersection", JCGT 2(1), section 3), arithmetic
 * contract of the header:
 *   kz = axis of d's largest magnitude (ties: x before y before z), kx = kz + 1, ky = kx + 1 (mod 3; no winding swap: nothing is culled by
 *   orientation);  sz = 1 / d[kz],  sx = d[kx] * sz,  sy = d[ky] * sz;
 *   per vertex P:  z = P[kz] - o[kz],  x = fma(-sx, z, P[kx] - o[kx]),  y = fma(-sy, z, P[ky] - o[ky]);
 *   U = Cx*By - Cy*Bx, V = Ax*Cy - Ay*Cx, W = Bx*Ay - By*Ax with plain products and a plain difference; if any of them is 0 all three are
 *   re-evaluated in double and rounded to float;  det = (U + V) + W;  t = (fma(U, Az, fma(V, Bz, W*Cz)) * sz) * (1/det);
 *   (b1, b2) = (V, W) * (1/det);  accept iff no two of U, V, W have strictly opposite signs and 0 <= t < inf.
 * Every vertex is mapped by the same function of (vertex, ray) in every triangle it belongs to; rnd(a*b) - rnd(c*d) is exactly antisymmetric
 * in the two vertices and never has the opposite sign of the exact value, and a zero gets its exact sign from the double evaluation: shared
 * edges and shared vertices cannot leak. */
typedef struct { float ox, oy, oz, sx, sy, sz; int kx, ky, kz; } ray_xf;
static inline float pick3(v3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
static inline ray_xf ray_xform(v3 o, v3 d) {
    ray_xf x;
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    x.kz = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    x.kx = x.kz == 2 ? 0 : x.kz + 1; x.ky = x.kx == 2 ? 0 : x.kx + 1;
    x.sz = 1.0f / pick3(d, x.kz);
    x.sx = pick3(d, x.kx) * x.sz; x.sy = pick3(d, x.ky) * x.sz;
    x.ox = pick3(o, x.kx); x.oy = pick3(o, x.ky); x.oz = pick3(o, x.kz);
    return x;
}
static inline int tri_test(const orc_scene *sc, int64_t f, const ray_xf *x, float *t_, float *u_, float *v_) {
    const int32_t *fi = sc->faces + f * 3;
    v3 p0 = v3_ld(sc->verts + (int64_t)fi[0] * 3), p1 = v3_ld(sc->verts + (int64_t)fi[1] * 3), p2 = v3_ld(sc->verts + (int64_t)fi[2] * 3);
    float Atz = pick3(p0, x->kz) - x->oz, Btz = pick3(p1, x->kz) - x->oz, Ctz = pick3(p2, x->kz) - x->oz;
    float Ax = fmaf(-x->sx, Atz, pick3(p0, x->kx) - x->ox), Ay = fmaf(-x->sy, Atz, pick3(p0, x->ky) - x->oy);
    float Bx = fmaf(-x->sx, Btz, pick3(p1, x->kx) - x->ox), By = fmaf(-x->sy, Btz, pick3(p1, x->ky) - x->oy);
    float Cx = fmaf(-x->sx, Ctz, pick3(p2, x->kx) - x->ox), Cy = fmaf(-x->sy, Ctz, pick3(p2, x->ky) - x->oy);
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (fminf(fminf(fabsf(U), fabsf(V)), fabsf(W)) == 0.f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    float det = (U + V) + W;
    float inv_det = 1.0f / det;
    float t = (fmaf(U, Atz, fmaf(V, Btz, W * Ctz)) * x->sz) * inv_det;
    float u = V * inv_det, v = W * inv_det;      /* (b1, b2): p = b0 p0 + b1 p1 + b2 p2 */
    float mn = fminf(fminf(U, V), W), mx = fmaxf(fmaxf(U, V), W);
    if (!(mn < 0.f && mx > 0.f) && t >= 0.f && t < INFINITY) { *t_ = t; *u_ = u; *v_ = v; return 1; }   /* det == 0: t is NaN or +-inf */
    return 0;
}
static inline void hit_update(orc_hit *h, float t, float u, float v, int64_t f) {
    if (t < h->t || (t == h->t && f < h->tri)) { h->t = t; h->u = u; h->v = v; h->tri = f; }
}

static orc_hit intersect_brute(const orc_scene *sc, v3 o, v3 d) {
    orc_hit h = {INFINITY, 0.f, 0.f, -1};
    const ray_xf xf = ray_xform(o, d);
    for (int64_t f = 0; f < sc->nf; ++f) {
        float t, u, v;
        if (tri_test(sc, f, &xf, &t, &u, &v)) { if (h.tri < 0) { h.t = t; h.u = u; h.v = v; h.tri = f; } else hit_update(&h, t, u, v, f); }
    }
    return h;
}

/* ---- a small binned-SAH BVH2 (oracle's own; only an accelerator for the brute-force semantics) ---- */
typedef struct { float lo[3], hi[3], c[3]; } tri_box;
typedef struct { orc_scene *sc; tri_box *tb; float pad; } build_ctx;

static void box_reset(float *lo, float *hi) { for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; } }
static void box_grow(float *lo, float *hi, const float *blo, const float *bhi) {
    for (int k = 0; k < 3; ++k) { if (blo[k] < lo[k]) lo[k] = blo[k]; if (bhi[k] > hi[k]) hi[k] = bhi[k]; }
}
static float box_area(const float *lo, const float *hi) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0) return 0.f;
    return 2.f * (dx * dy + dy * dz + dz * dx);
}
#define ORC_BINS 16
#define ORC_LEAF 4
static int32_t build_rec(build_ctx *cx, int32_t start, int32_t count) {
    orc_scene *sc = cx->sc;
    int32_t me = sc->n_nodes++;
    orc_node *nd = &sc->nodes[me];
    float lo[3], hi[3], clo[3], chi[3];
    box_reset(lo, hi); box_reset(clo, chi);
    for (int32_t i = start; i < start + count; ++i) {
        tri_box *b = &cx->tb[sc->order[i]];
        box_grow(lo, hi, b->lo, b->hi); box_grow(clo, chi, b->c, b->c);
    }
    for (int k = 0; k < 3; ++k) { nd->lo[k] = lo[k] - cx->pad; nd->hi[k] = hi[k] + cx->pad; }
    nd->left = -1; nd->start = start; nd->count = count;
    if (count <= ORC_LEAF) return me;
    /* binned SAH over the widest centroid axis candidates */
    int best_axis = -1, best_bin = -1; float best_cost = INFINITY;
    for (int ax = 0; ax < 3; ++ax) {
        float ext = chi[ax] - clo[ax];
        if (!(ext > 0.f)) continue;
        float blo[ORC_BINS][3], bhi[ORC_BINS][3]; int bc[ORC_BINS];
        for (int b = 0; b < ORC_BINS; ++b) { box_reset(blo[b], bhi[b]); bc[b] = 0; }
        float sc_ = (float)ORC_BINS / ext;
        for (int32_t i = start; i < start + count; ++i) {
            tri_box *t = &cx->tb[sc->order[i]];
            int b = (int)((t->c[ax] - clo[ax]) * sc_); if (b >= ORC_BINS) b = ORC_BINS - 1; if (b < 0) b = 0;
            box_grow(blo[b], bhi[b], t->lo, t->hi); bc[b]++;
        }
        float ra[ORC_BINS]; int rc[ORC_BINS];
        float alo[3], ahi[3]; int n = 0; box_reset(alo, ahi);
        for (int b = ORC_BINS - 1; b > 0; --b) { box_grow(alo, ahi, blo[b], bhi[b]); n += bc[b]; ra[b] = box_area(alo, ahi); rc[b] = n; }
        box_reset(alo, ahi); n = 0;
        for (int b = 0; b < ORC_BINS - 1; ++b) {
            box_grow(alo, ahi, blo[b], bhi[b]); n += bc[b];
            if (n == 0 || rc[b + 1] == 0) continue;
            float cost = box_area(alo, ahi) * (float)n + ra[b + 1] * (float)rc[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
        }
    }
    int32_t mid;
    if (best_axis < 0) {
        mid = start + count / 2; /* all centroids coincide */
    } else {
        float ext = chi[best_axis] - clo[best_axis], sc_ = (float)ORC_BINS / ext;
        int32_t i = start, j = start + count - 1;
        while (i <= j) {
            tri_box *t = &cx->tb[sc->order[i]];
            int b = (int)((t->c[best_axis] - clo[best_axis]) * sc_); if (b >= ORC_BINS) b = ORC_BINS - 1; if (b < 0) b = 0;
            if (b <= best_bin) ++i; else { int32_t tmp = sc->order[i]; sc->order[i] = sc->order[j]; sc->order[j] = tmp; --j; }
        }
        mid = i;
        if (mid == start || mid == start + count) mid = start + count / 2;
    }
    int32_t l = build_rec(cx, start, mid - start);
    int32_t r = build_rec(cx, mid, start + count - mid);
    nd = &sc->nodes[me];
    nd->left = l; nd->start = r; /* for internal nodes `start` holds the right child */
    nd->count = 0;
    return me;
}

ORC_API orc_scene *orc_scene_create(const float *verts, int64_t nv, const int32_t *faces, int64_t nf) {
    orc_scene *sc = (orc_scene *)calloc(1, sizeof(orc_scene));
    sc->nv = nv; sc->nf = nf; sc->verts = verts; sc->faces = faces;
    if (nf == 0) return sc;
    sc->nodes = (orc_node *)malloc(sizeof(orc_node) * (size_t)(2 * nf));
    sc->order = (int32_t *)malloc(sizeof(int32_t) * (size_t)nf);
    tri_box *tb = (tri_box *)malloc(sizeof(tri_box) * (size_t)nf);
    float glo[3], ghi[3]; box_reset(glo, ghi);
    for (int64_t f = 0; f < nf; ++f) {
        sc->order[f] = (int32_t)f;
        box_reset(tb[f].lo, tb[f].hi);
        for (int k = 0; k < 3; ++k) {
            const float *p = verts + (int64_t)faces[f * 3 + k] * 3;
            box_grow(tb[f].lo, tb[f].hi, p, p);
        }
        for (int k = 0; k < 3; ++k) tb[f].c[k] = 0.5f * (tb[f].lo[k] + tb[f].hi[k]);
        box_grow(glo, ghi, tb[f].lo, tb[f].hi);
    }
    float ext = 0.f;
    for (int k = 0; k < 3; ++k) { float e = ghi[k] - glo[k]; if (e > ext) ext = e; float a = fabsf(glo[k]), b = fabsf(ghi[k]); if (a > ext) ext = a; if (b > ext) ext = b; }
    build_ctx cx = {sc, tb, 1e-4f * ext + 1e-30f};
    build_rec(&cx, 0, (int32_t)nf);
    free(tb);
    return sc;
}
ORC_API void orc_scene_destroy(orc_scene *sc) { if (sc) { free(sc->nodes); free(sc->order); free(sc); } }

static inline int slab(const orc_node *n, v3 o, v3 id, float tbest, float *tnear) {
    float t0 = (n->lo[0] - o.x) * id.x, t1 = (n->hi[0] - o.x) * id.x;
    float tmin = fminf(t0, t1), tmax = fmaxf(t0, t1);
    t0 = (n->lo[1] - o.y) * id.y; t1 = (n->hi[1] - o.y) * id.y;
    tmin = fmaxf(tmin, fminf(t0, t1)); tmax = fminf(tmax, fmaxf(t0, t1));
    t0 = (n->lo[2] - o.z) * id.z; t1 = (n->hi[2] - o.z) * id.z;
    tmin = fmaxf(tmin, fminf(t0, t1)); tmax = fminf(tmax, fmaxf(t0, t1));
    *tnear = tmin;
    /* conservative: a NaN slab (0*inf) is ignored by fmin/fmax; generous relative slack on both ends */
    return tmax >= 0.f && tmin <= tmax * 1.00001f + 1e-30f && tmin * 0.9999f <= tbest;
}
static inline float safe_inv(float d) {
    if (fabsf(d) < 1e-30f) return d < 0.f || (d == 0.f && signbit(d)) ? -1e30f : 1e30f;
    return 1.0f / d;
}
/* ordered (near child first) stack traversal; only an accelerator for the brute-force semantics */
static orc_hit intersect_bvh(const orc_scene *sc, v3 o, v3 d, int64_t *n_nodes, int64_t *n_tris) {
    orc_hit h = {INFINITY, 0.f, 0.f, -1};
    if (sc->nf == 0) return h;
    v3 id = v3_make(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    const ray_xf xf = ray_xform(o, d);
    int32_t stack[128]; int sp = 0;
    int32_t cur = 0; float tn;
    if (n_nodes) ++*n_nodes;
    if (!slab(&sc->nodes[0], o, id, h.t, &tn)) return h;
    for (;;) {
        const orc_node *n = &sc->nodes[cur];
        if (n->left < 0) {
            for (int32_t i = n->start; i < n->start + n->count; ++i) {
                float t, u, v; int64_t f = sc->order[i];
                if (n_tris) ++*n_tris;
                if (tri_test(sc, f, &xf, &t, &u, &v)) { if (h.tri < 0) { h.t = t; h.u = u; h.v = v; h.tri = f; } else hit_update(&h, t, u, v, f); }
            }
        } else {
            float tl, tr;
            if (n_nodes) *n_nodes += 2;
            int hl = slab(&sc->nodes[n->left], o, id, h.t, &tl), hr = slab(&sc->nodes[n->start], o, id, h.t, &tr);
            if (hl && hr) {
                if (tl <= tr) { stack[sp++] = n->start; cur = n->left; } else { stack[sp++] = n->left; cur = n->start; }
                continue;
            }
            if (hl) { cur = n->left; continue; }
            if (hr) { cur = n->start; continue; }
        }
        /* pop; entries were valid when pushed, re-check against the current best */
        for (;;) {
            if (!sp) return h;
            cur = stack[--sp];
            if (n_nodes) ++*n_nodes;
            if (slab(&sc->nodes[cur], o, id, h.t, &tn)) break;
        }
    }
}

/* Mitsuba Mesh::compute_surface_interaction (restated): p = fma(p0,b0,fma(p1,b1,p2*b2)), b0 = 1-b1-b2 */
static inline v3 hit_position(const orc_scene *sc, const orc_hit *h) {
    const int32_t *fi = sc->faces + h->tri * 3;
    const float *p0 = sc->verts + (int64_t)fi[0] * 3, *p1 = sc->verts + (int64_t)fi[1] * 3, *p2 = sc->verts + (int64_t)fi[2] * 3;
    float b1 = h->u, b2 = h->v, b0 = (1.f - b1) - b2;
    return v3_make(fmaf(p0[0], b0, fmaf(p1[0], b1, p2[0] * b2)), fmaf(p0[1], b0, fmaf(p1[1], b1, p2[1] * b2)),
                   fmaf(p0[2], b0, fmaf(p1[2], b1, p2[2] * b2)));
}
static inline v3 hit_normal(const orc_scene *sc, const orc_hit *h) {
    const int32_t *fi = sc->faces + h->tri * 3;
    v3 p0 = v3_ld(sc->verts + (int64_t)fi[0] * 3), p1 = v3_ld(sc->verts + (int64_t)fi[1] * 3), p2 = v3_ld(sc->verts + (int64_t)fi[2] * 3);
    v3 n = x_cross(v3_sub(p1, p0), v3_sub(p2, p0));
    float len = sqrtf(x_dot(n, n));
    return v3_make(n.x / len, n.y / len, n.z / len); /* si.n = normalize(cross(dp0,dp1)) */
}

/* utils/path_tracing.py:17-48 ray_intersect.  mode 0 = brute force, 1 = BVH.
 * Miss: idx=-1, valid=0, positions/normals/uv = 0 (the reference leaves Mitsuba's values there; callers mask by valid). */
ORC_API void orc_ray_intersect(const orc_scene *sc, const float *xs, const float *ds, int64_t B, int mode,
                               float *pos, float *nrm, float *uv, int64_t *idx, uint8_t *valid, float *t_out, int64_t *counters) {
    int64_t tot_nodes = 0, tot_tris = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : tot_nodes, tot_tris)
    for (int64_t i = 0; i < B; ++i) {
        v3 o = v3_ld(xs + i * 3), d = v3_ld(ds + i * 3);
        int64_t nn = 0, nt = 0;
        orc_hit h = mode == 0 ? intersect_brute(sc, o, d) : intersect_bvh(sc, o, d, &nn, &nt);
        tot_nodes += nn; tot_tris += nt;
        if (h.tri >= 0) {
            if (pos) v3_st(pos + i * 3, hit_position(sc, &h));
            if (nrm) {
                v3 n = t_normalize(hit_normal(sc, &h));         /* NF.normalize(ret.n) */
                v3 mv = v3_make(-d.x, -d.y, -d.z);               /* double_sided(-ds, normals) */
                if (t_dot(n, mv) < 0.f) n = v3_make(-n.x, -n.y, -n.z);
                v3_st(nrm + i * 3, n);
            }
            if (uv) { uv[i * 2] = h.u; uv[i * 2 + 1] = h.v; }
            if (idx) idx[i] = h.tri;
            if (valid) valid[i] = 1;
            if (t_out) t_out[i] = h.t;
        } else {
            if (pos) { pos[i * 3] = pos[i * 3 + 1] = pos[i * 3 + 2] = 0.f; }
            if (nrm) { nrm[i * 3] = nrm[i * 3 + 1] = nrm[i * 3 + 2] = 0.f; }
            if (uv) { uv[i * 2] = uv[i * 2 + 1] = 0.f; }
            if (idx) idx[i] = -1;
            if (valid) valid[i] = 0;
            if (t_out) t_out[i] = INFINITY;
        }
    }
    if (counters) { counters[0] = tot_nodes; counters[1] = tot_tris; }
}

/* ============================================================================================
 * Philox4x32-10 counter RNG (perf-mode uniforms; integer work, must match the HIP kernel bit for bit)
 *   counter = (idx_lo, idx_hi, stream, 0), key = (seed_lo, seed_hi); u0=(c0>>8)*2^-24, u1=(c1>>8)*2^-24
 * ========================================================================================== */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        if (r) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
}
static inline void philox_u2(uint64_t seed, uint64_t idx, uint32_t stream, float *u0, float *u1) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), stream, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    *u0 = (float)(c[0] >> 8) * 5.9604644775390625e-08f;
    *u1 = (float)(c[1] >> 8) * 5.9604644775390625e-08f;
}
ORC_API void orc_philox_u2(uint64_t seed, uint64_t idx0, uint32_t stream, int64_t n, float *u2) {
    for (int64_t i = 0; i < n; ++i) philox_u2(seed, idx0 + (uint64_t)i, stream, u2 + i * 2, u2 + i * 2 + 1);
}

/* ============================================================================================
 * a6/a7  the bake loop body  (bake_shading.py:108-123 diffuse, :168-188 specular)
 *
 * pos/nrm/wo: (P,3) primary-hit records of the valid pixels.  u2: (P*spp,2) explicit uniforms in the
 * reference's order (row = pixel*spp + sample: concatenation of the per-chunk torch.rand draws), or
 * NULL -> Philox keyed by (seed, pix_id[p]*spp+s, stream).  rough<0 -> diffuse lobe.
 * RayEpsilon = 1500 * 2^-24 (mitsuba.math.RayEpsilon for float32).
 * Outputs: out0 (P,3) = Ld or Ls0, out1 (P,3) = Ls1 (specular only); tri_next (P*spp) optional.
 * ========================================================================================== */
static const float RAY_EPS = 1500.0f * 5.9604644775390625e-08f;

/* src_next (P*spp, optional): per sample, the radiance-table row eval_emitter read -- -2 - emitter ordinal for an emitter triangle,
 * the VoxelSLF row for the radiance cache, -1 for empty space or a miss.  Parity bookkeeping: a sample whose (tri_next, src_next)
 * differs between two evaluations of the same path is a discrete "flip"; all other differences are rounding. */
ORC_API void orc_bake_src(const orc_scene *sc, const orc_emitter *em, const orc_slf *slf,
                          const float *pos, const float *nrm, const float *wo, int64_t P, int spp,
                          const float *u2, uint64_t seed, uint32_t stream, const int32_t *pix_id, float rough,
                          float *out0, float *out1, int64_t *tri_next, int64_t *src_next, int64_t *counters);
ORC_API void orc_bake(const orc_scene *sc, const orc_emitter *em, const orc_slf *slf,
                      const float *pos, const float *nrm, const float *wo, int64_t P, int spp,
                      const float *u2, uint64_t seed, uint32_t stream, const int32_t *pix_id, float rough,
                      float *out0, float *out1, int64_t *tri_next, int64_t *counters) {
    orc_bake_src(sc, em, slf, pos, nrm, wo, P, spp, u2, seed, stream, pix_id, rough, out0, out1, tri_next, NULL, counters);
}
ORC_API void orc_bake_src(const orc_scene *sc, const orc_emitter *em, const orc_slf *slf,
                          const float *pos, const float *nrm, const float *wo, int64_t P, int spp,
                          const float *u2, uint64_t seed, uint32_t stream, const int32_t *pix_id, float rough,
                          float *out0, float *out1, int64_t *tri_next, int64_t *src_next, int64_t *counters) {
    int64_t tot_nodes = 0, tot_tris = 0;
    const int specular = rough >= 0.f;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : tot_nodes, tot_tris)
    for (int64_t p = 0; p < P; ++p) {
        v3 x = v3_ld(pos + p * 3), n = v3_ld(nrm + p * 3);
        v3 w = specular ? v3_ld(wo + p * 3) : v3_make(0, 0, 0);
        double a0[3] = {0, 0, 0}, a1[3] = {0, 0, 0};
        /* mode 1: the kernels' reduction -- lane l accumulates samples l, l+L, l+2L, ... in f32 (L = lanes per pixel),
           then an xor butterfly over the L lanes, then * (1.0f/spp) */
        float lane0[64][3], lane1[64][3];
        int L = 64;
        if (spp < 64 && (spp & (spp - 1)) == 0) L = spp;
        if (g_mode == 1) { memset(lane0, 0, sizeof(lane0)); memset(lane1, 0, sizeof(lane1)); }
        for (int s = 0; s < spp; ++s) {
            float u0, u1;
            if (u2) { u0 = u2[(p * spp + s) * 2]; u1 = u2[(p * spp + s) * 2 + 1]; }
            else philox_u2(seed, (uint64_t)(pix_id ? pix_id[p] : p) * (uint64_t)spp + (uint64_t)s, stream, &u0, &u1);
            v3 wi; float g0 = 1.f, g1 = 0.f;
            if (specular) { spec_sample r = sample_specular1(u0, u1, w, n, rough); wi = r.wi; g0 = r.g0; g1 = r.g1; }
            else wi = diffuse_sampler(u0, u1, n);
            /* position + RayEpsilon*wi (bake_shading.py:117,180) */
            v3 o = v3_make(x.x + RAY_EPS * wi.x, x.y + RAY_EPS * wi.y, x.z + RAY_EPS * wi.z);
            int64_t nn = 0, nt = 0;
            orc_hit h = intersect_bvh(sc, o, wi, &nn, &nt);
            tot_nodes += nn; tot_tris += nt;
            if (tri_next) tri_next[p * spp + s] = h.tri;
            v3 pn = h.tri >= 0 ? hit_position(sc, &h) : v3_make(0, 0, 0);
            /* eval_emitter(p_next, wi, tri_next, ones, trace_roughness=0.0) (bake_shading.py:121-122) */
            v3 Le = eval_emitter1(em, slf, pn, h.tri, 1, 1.0f, 0.0f, NULL, NULL);
            if (src_next) {   /* restates eval_emitter1's table choice (model/emitter.py:196-216) */
                int64_t src = -1;
                if (h.tri >= 0) src = em->is_emitter[h.tri] ? -2 - em->emitter_idx[h.tri] : slf_spatial_idx(slf, pn);
                src_next[p * spp + s] = src;
            }
            if (g_mode == 1) {
                const int l = s % L;
                if (specular) {
                    lane0[l][0] += Le.x * g0; lane0[l][1] += Le.y * g0; lane0[l][2] += Le.z * g0;
                    lane1[l][0] += Le.x * g1; lane1[l][1] += Le.y * g1; lane1[l][2] += Le.z * g1;
                } else { lane0[l][0] += Le.x; lane0[l][1] += Le.y; lane0[l][2] += Le.z; }
            } else if (specular) {
                a0[0] += (double)(Le.x * g0); a0[1] += (double)(Le.y * g0); a0[2] += (double)(Le.z * g0);
                a1[0] += (double)(Le.x * g1); a1[1] += (double)(Le.y * g1); a1[2] += (double)(Le.z * g1);
            } else { a0[0] += Le.x; a0[1] += Le.y; a0[2] += Le.z; }
        }
        if (g_mode == 1) {
            for (int m = 1; m < L; m <<= 1) {
                float t0[64][3], t1[64][3];
                for (int l = 0; l < L; ++l) for (int c = 0; c < 3; ++c) { t0[l][c] = lane0[l][c] + lane0[l ^ m][c]; t1[l][c] = lane1[l][c] + lane1[l ^ m][c]; }
                memcpy(lane0, t0, sizeof(float) * 3 * L); memcpy(lane1, t1, sizeof(float) * 3 * L);
            }
            const float inv_spp = 1.0f / (float)spp;
            for (int c = 0; c < 3; ++c) {
                out0[p * 3 + c] = lane0[0][c] * inv_spp;
                if (specular && out1) out1[p * 3 + c] = lane1[0][c] * inv_spp;
            }
            continue;
        }
        /* .reshape(b,spp,3).mean(1): summation order of torch's reduction is unspecified; we accumulate in
           double and round once (differs from any f32 order by <= spp*2^-24 relative) */
        for (int c = 0; c < 3; ++c) {
            out0[p * 3 + c] = (float)(a0[c] / (double)spp);
            if (specular && out1) out1[p * 3 + c] = (float)(a1[c] / (double)spp);
        }
    }
    if (counters) { counters[0] = tot_nodes; counters[1] = tot_tris; }
}

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ============================================================================================
 * a9 (cfg 5)  path_tracing_single  (utils/path_tracing.py:320-407) and its building blocks
 * ========================================================================================== */
typedef struct { v3 albedo; float rough, metal; } orc_mat;

/* model/brdf.py:138-175 eval_brdf */
static inline void eval_brdf1(v3 wi, v3 wo, v3 n, orc_mat m, v3 *brdf, float *pdf) {
    v3 h = t_normalize(v3_make(wi.x + wo.x, wi.y + wo.y, wi.z + wo.z));
    float NoL = relu(t_dot(wi, n)), NoV = relu(t_dot(wo, n));
    float VoH = relu(t_dot(wo, h)), NoH = relu(t_dot(n, h));
    float D = D_GGX(NoH, m.rough);
    float vc = VoH < 1e-4f ? 1e-4f : VoH;
    float pdf_spec = D / (4.f * vc) * NoH;
    float pdf_diff = NoL / PI_F;
    *pdf = 0.5f * pdf_spec + 0.5f * pdf_diff;
    float om = 1.f - m.metal;
    v3 kd = v3_make(m.albedo.x * om, m.albedo.y * om, m.albedo.z * om);
    v3 ks = v3_make(0.04f * om + m.albedo.x * m.metal, 0.04f * om + m.albedo.y * m.metal, 0.04f * om + m.albedo.z * m.metal);
    float G = G1_GGX_Schlick(NoL, m.rough) * G1_GGX_Schlick(NoV, m.rough);
    float x = pow5(1.f - VoH);
    float dg = D * G;
    brdf->x = kd.x / PI_F * NoL + dg * (ks.x + (1.f - ks.x) * x) / 4.0f * NoL;
    brdf->y = kd.y / PI_F * NoL + dg * (ks.y + (1.f - ks.y) * x) / 4.0f * NoL;
    brdf->z = kd.z / PI_F * NoL + dg * (ks.z + (1.f - ks.z) * x) / 4.0f * NoL;
}
/* model/brdf.py:177-210 sample_brdf */
static inline void sample_brdf1(float s1, float u0, float u1, v3 wo, v3 n, orc_mat m, v3 *wi, float *pdf, v3 *w) {
    *wi = (s1 > 0.5f) ? diffuse_sampler(u0, u1, n) : specular_sampler(u0, u1, m.rough, wo, n);
    v3 brdf;
    eval_brdf1(*wi, wo, n, m, &brdf, pdf);
    *w = v3_make(0.f, 0.f, 0.f);
    if (*pdf > 0.f) {
        *w = v3_make(brdf.x / *pdf, brdf.y / *pdf, brdf.z / *pdf);
        if (w->x != w->x) w->x = 0.f;
        if (w->y != w->y) w->y = 0.f;
        if (w->z != w->z) w->z = 0.f;
    }
}
/* model/emitter.py:224-255 sample_emitter */
static inline void sample_emitter1(const orc_emitter *e, float s1, float u0, float u1, v3 pos, v3 *wi, float *pdf, int64_t *tri) {
    const float v = s1 < 1e-12f ? 1e-12f : s1;
    int64_t lo = 0, hi = e->k;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (e->cdf[mid] < v) lo = mid + 1; else hi = mid; }
    const int64_t ei = lo < e->k ? lo : e->k - 1;
    const float xi1 = sqrtf(u0);
    const float u = 1.f - xi1, vv = xi1 * u1, w = (1.f - u) - vv;
    const float *p = e->verts + ei * 9;
    v3 p1 = v3_make((p[0] * u + p[3] * vv) + p[6] * w, (p[1] * u + p[4] * vv) + p[7] * w, (p[2] * u + p[5] * vv) + p[8] * w);
    *wi = t_normalize(v3_sub(p1, pos));
    float a = e->area[ei]; if (a < 1e-12f) a = 1e-12f;
    *pdf = e->emitter_pdf / a;
    *tri = e->ord2tri[ei];
}
static inline orc_mat mat_at(const float *albedo, const float *rough, const float *metal, int64_t i) {
    orc_mat m; m.albedo = v3_ld(albedo + i * 3); m.rough = rough[i]; m.metal = metal[i];
    return m;
}
ORC_API void orc_sample_emitter(const orc_emitter *e, const float *s1, const float *s2, const float *pos, int64_t N, float *wi, float *pdf, int64_t *tri) {
    for (int64_t i = 0; i < N; ++i) { v3 w; sample_emitter1(e, s1[i], s2[i * 2], s2[i * 2 + 1], v3_ld(pos + i * 3), &w, pdf + i, tri + i); v3_st(wi + i * 3, w); }
}
ORC_API void orc_eval_brdf(const float *wi, const float *wo, const float *nrm, const float *albedo, const float *rough, const float *metal, int64_t N,
                           float *brdf, float *pdf) {
    for (int64_t i = 0; i < N; ++i) { v3 b; eval_brdf1(v3_ld(wi + i * 3), v3_ld(wo + i * 3), v3_ld(nrm + i * 3), mat_at(albedo, rough, metal, i), &b, pdf + i); v3_st(brdf + i * 3, b); }
}
ORC_API void orc_sample_brdf(const float *s1, const float *s2, const float *wo, const float *nrm, const float *albedo, const float *rough, const float *metal,
                             int64_t N, float *wi, float *pdf, float *weight) {
    for (int64_t i = 0; i < N; ++i) {
        v3 w, bw;
        sample_brdf1(s1[i], s2[i * 2], s2[i * 2 + 1], v3_ld(wo + i * 3), v3_ld(nrm + i * 3), mat_at(albedo, rough, metal, i), &w, pdf + i, &bw);
        v3_st(wi + i * 3, w); v3_st(weight + i * 3, bw);
    }
}
/* utils/path_tracing.py:338-340 */
ORC_API void orc_pt_jitter(const float *rays_d, const float *dxdu, const float *dydv, const float *dudv, int64_t B, int spp, float *wi) {
    const int64_t n = B * spp;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t b = i / spp;
        const float du = dudv[i] - 0.5f, dv = dudv[n + i] - 0.5f;
        v3 d = v3_ld(rays_d + b * 3), dx = v3_ld(dxdu + b * 3), dy = v3_ld(dydv + b * 3);
        v3_st(wi + i * 3, t_normalize(v3_make((d.x + dx.x * du) + dy.x * dv, (d.y + dx.y * du) + dy.y * dv, (d.z + dx.z * du) + dy.z * dv)));
    }
}
static inline v3 ff_normal(const orc_scene *sc, const orc_hit *h, v3 d) {
    v3 n = t_normalize(hit_normal(sc, h));
    if (t_dot(n, v3_make(-d.x, -d.y, -d.z)) < 0.f) n = v3_make(-n.x, -n.y, -n.z);
    return n;
}
/* :357-382  term1 = coef1 * radiance[e1] */
ORC_API void orc_pt_nee(const orc_scene *sc, const orc_emitter *e, const float *pos, const float *nrm, const float *wo, const float *albedo,
                        const float *rough, const float *metal, const float *s1, const float *s2, int64_t N, float *coef1, int32_t *e1,
                        float g_eps, float pdf_eps, float mis_eps) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i) {
        const v3 x = v3_ld(pos + i * 3), n = v3_ld(nrm + i * 3), w = v3_ld(wo + i * 3);
        v3 wi; float emit_pdf; int64_t emit_tri;
        sample_emitter1(e, s1[i], s2[i * 2], s2[i * 2 + 1], x, &wi, &emit_pdf, &emit_tri);
        const v3 o = v3_make(x.x + RAY_EPS * wi.x, x.y + RAY_EPS * wi.y, x.z + RAY_EPS * wi.z);
        orc_hit h = intersect_bvh(sc, o, wi, NULL, NULL);
        const int emit_valid = h.tri >= 0;
        int64_t ord = -1; float G = 1.f; int emit_vis = 1;
        if (emit_valid) {
            const v3 ep = hit_position(sc, &h), en = ff_normal(sc, &h, wi);
            emit_vis = emit_tri == h.tri;
            ord = e->is_emitter[h.tri] ? e->emitter_idx[h.tri] : -1;
            const v3 dl = v3_sub(ep, x);
            float d2 = (dl.x * dl.x + dl.y * dl.y) + dl.z * dl.z; if (d2 < g_eps) d2 = g_eps;
            G = fabsf(t_dot(v3_make(-wi.x, -wi.y, -wi.z), en)) / d2;
        }
        v3 brdf; float brdf_pdf;
        eval_brdf1(wi, w, n, mat_at(albedo, rough, metal, i), &brdf, &brdf_pdf);
        brdf_pdf = brdf_pdf * G;
        float w_mis = 0.f;
        if (emit_pdf > 0.f && !isinf(brdf_pdf)) { float den = emit_pdf * emit_pdf + brdf_pdf * brdf_pdf; if (mis_eps > 0.f && den < mis_eps) den = mis_eps; w_mis = emit_pdf * emit_pdf / den; }
        if (isinf(emit_pdf) || brdf_pdf == 0.f) w_mis = 1.f;
        const float sv = emit_vis ? 1.f : 0.f;
        const float ew = G / (emit_pdf < pdf_eps ? pdf_eps : emit_pdf);
        v3_st(coef1 + i * 3, v3_make(brdf.x * (sv * ew) * w_mis, brdf.y * (sv * ew) * w_mis, brdf.z * (sv * ew) * w_mis));
        e1[i] = (emit_valid && ord >= 0) ? (int32_t)ord : -1;
    }
}
/* :384-391 */
ORC_API void orc_pt_brdf_trace(const orc_scene *sc, const float *pos, const float *nrm, const float *wo, const float *albedo, const float *rough,
                               const float *metal, const float *s1, const float *s2, int64_t N, float *wi_out, float *pdf_out, float *w_out,
                               float *pos_next, float *nrm_next, int64_t *tri_next, uint8_t *valid, int lobe, float lobe_rough) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i) {
        const v3 x = v3_ld(pos + i * 3), n = v3_ld(nrm + i * 3), w = v3_ld(wo + i * 3);
        v3 wi, bw; float pdf;
        if (lobe == 0) sample_brdf1(s1[i], s2[i * 2], s2[i * 2 + 1], w, n, mat_at(albedo, rough, metal, i), &wi, &pdf, &bw);
        else if (lobe == 1) { wi = diffuse_sampler(s2[i * 2], s2[i * 2 + 1], n); pdf = relu(t_dot(n, wi)) / PI_F; bw = v3_make(1.f, 1.f, 1.f); }   /* model/brdf.py:78-88 */
        else { spec_sample r = sample_specular1(s2[i * 2], s2[i * 2 + 1], w, n, lobe_rough); wi = r.wi; pdf = r.pdf; bw = v3_make(r.g0, r.g1, 0.f); }  /* :112-136 */
        const v3 o = v3_make(x.x + RAY_EPS * wi.x, x.y + RAY_EPS * wi.y, x.z + RAY_EPS * wi.z);
        orc_hit h = intersect_bvh(sc, o, wi, NULL, NULL);
        v3 pn = v3_make(0, 0, 0), nn = v3_make(0, 0, 0);
        if (h.tri >= 0) { pn = hit_position(sc, &h); nn = ff_normal(sc, &h, wi); }
        v3_st(wi_out + i * 3, wi); pdf_out[i] = pdf; v3_st(w_out + i * 3, bw);
        v3_st(pos_next + i * 3, pn); v3_st(nrm_next + i * 3, nn); tri_next[i] = h.tri; valid[i] = h.tri >= 0;
    }
}
/* :394-404  term2 = coef2 * radiance[e2] + const2 */
ORC_API void orc_pt_brdf_finish(const orc_emitter *e, const orc_slf *slf, const float *pos, const float *pos_next, const float *nrm_next, const float *wi_in,
                                const int64_t *tri_next, const float *rough_next, const float *pdf_in, const float *w_in, int64_t N, float *coef2,
                                float *const2, int32_t *e2, uint8_t *valid_next_out, float trace_rough, float g_eps) {
    for (int64_t i = 0; i < N; ++i) {
        const v3 x = v3_ld(pos + i * 3), pn = v3_ld(pos_next + i * 3), nn = v3_ld(nrm_next + i * 3), wi = v3_ld(wi_in + i * 3);
        const int64_t tri = tri_next[i];
        const int vis = tri != -1;
        const int is_area = vis && e->is_emitter[tri];
        int64_t ord = is_area ? e->emitter_idx[tri] : -1;
        float emit_pdf = 0.f;
        if (is_area) { float a = e->area[ord]; if (a < 1e-12f) a = 1e-12f; emit_pdf = e->emitter_pdf / a; }
        int valid_next = (!is_area) && vis;
        v3 sl = v3_make(0, 0, 0);
        if ((!is_area) && vis && rough_next[i] > trace_rough) { sl = slf_forward(slf, pn); if ((sl.x + sl.y) + sl.z > 0.f) valid_next = 0; }
        const v3 dl = v3_sub(x, pn);
        float d2 = (dl.x * dl.x + dl.y * dl.y) + dl.z * dl.z; if (d2 < g_eps) d2 = g_eps;
        float G = fabsf(t_dot(v3_make(-nn.x, -nn.y, -nn.z), wi)) / d2;
        if (!valid_next) G = 1.f;
        if (valid_next_out) valid_next_out[i] = (uint8_t)valid_next;
        const float brdf_pdf = pdf_in[i] * G;
        float w_mis = 0.f;
        if (brdf_pdf > 0.f && !isinf(emit_pdf)) w_mis = brdf_pdf * brdf_pdf / (emit_pdf * emit_pdf + brdf_pdf * brdf_pdf);
        if (isinf(brdf_pdf) || emit_pdf == 0.f) w_mis = 1.f;
        const v3 w = v3_ld(w_in + i * 3);
        v3_st(coef2 + i * 3, v3_make(w.x * w_mis, w.y * w_mis, w.z * w_mis));
        v3_st(const2 + i * 3, v3_make(w.x * sl.x * w_mis, w.y * sl.y * w_mis, w.z * sl.z * w_mis));
        e2[i] = is_area ? (int32_t)ord : -1;
    }
}
/* :344, :382, :404, :406  (same summation order as the HIP accumulate kernel) */
ORC_API void orc_pt_accumulate(const float *radiance, const int32_t *e0, const int32_t *path_of, const int32_t *e1, const float *coef1, const int32_t *e2,
                               const float *coef2, const float *const2, int64_t B, int spp, float *L) {
    for (int64_t b = 0; b < B; ++b) {
        float ax = 0.f, ay = 0.f, az = 0.f;
        for (int s = 0; s < spp; ++s) {
            const int64_t i = b * spp + s;
            v3 l = v3_make(0, 0, 0);
            if (e0[i] >= 0) l = v3_ld(radiance + (int64_t)e0[i] * 3);
            const int j = path_of[i];
            if (j >= 0) {
                if (e1[j] >= 0) { v3 r = v3_ld(radiance + (int64_t)e1[j] * 3), c = v3_ld(coef1 + (int64_t)j * 3); l.x += c.x * r.x; l.y += c.y * r.y; l.z += c.z * r.z; }
                v3 t2 = v3_ld(const2 + (int64_t)j * 3);
                if (e2[j] >= 0) { v3 r = v3_ld(radiance + (int64_t)e2[j] * 3), c = v3_ld(coef2 + (int64_t)j * 3); t2.x += c.x * r.x; t2.y += c.y * r.y; t2.z += c.z * r.z; }
                l.x += t2.x; l.y += t2.y; l.z += t2.z;
            }
            ax += l.x; ay += l.y; az += l.z;
        }
        const float inv = 1.0f / (float)spp;
        v3_st(L + b * 3, v3_make(ax * inv, ay * inv, az * inv));
    }
}
