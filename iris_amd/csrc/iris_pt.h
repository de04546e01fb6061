// cfg 5: the one-bounce MIS path tracer the reference back-propagates through
// (utils/path_tracing.py:320-407 path_tracing_single), with its building blocks SLFEmitter.sample_emitter
// (model/emitter.py:224-255), BaseBRDF.eval_brdf / sample_brdf (model/brdf.py:138-210).
//
// The material network (NGPBRDF, tiny-cuda-nn: third party) is evaluated by the caller between the stages, so the path is
// staged exactly where the reference calls material_net: jitter -> [ray_intersect, primary emitter] -> material ->
// {NEE stage, BRDF-sample stage} -> material -> finish stage -> accumulate.  Only emitter.radiance receives gradient
// (SURVEY.md section 3.4), and L is linear in it: every stage emits (emitter ordinal, rgb coefficient) pairs; the forward
// pass is a gather, the backward pass the matching scatter-add.
#pragma once
#include "iris_tile.h"

namespace iris {

struct EmitSampleDev {
    const float* cdf;       // (K) emitter_cdf exactly as torch computed it (model/emitter.py:170)
    const float* verts;     // (K,3,3) emitter_vertices
    const float* area;      // (K)
    const int32_t* ord2tri; // (K) triangle index of emitter ordinal (model/emitter.py:165-166)
    int64_t k;
    float emitter_pdf;      // 1/K
};

struct Mat { f3 albedo; float rough, metal; };

// model/brdf.py:138-175 eval_brdf
__device__ __forceinline__ void eval_brdf1(f3 wi, f3 wo, f3 n, Mat m, f3& brdf, float& pdf) {
    f3 h = t_normalize(mk3(wi.x + wo.x, wi.y + wo.y, wi.z + wo.z));
    float NoL = relu(t_dot(wi, n)), NoV = relu(t_dot(wo, n));
    float VoH = relu(t_dot(wo, h)), NoH = relu(t_dot(n, h));
    float D = D_GGX(NoH, m.rough);
    float pdf_spec = D / (4.f * fmaxf(VoH, 1e-4f)) * NoH;
    float pdf_diff = NoL / kPi;
    pdf = 0.5f * pdf_spec + 0.5f * pdf_diff;
    float om = 1.f - m.metal;
    f3 kd = mk3(m.albedo.x * om, m.albedo.y * om, m.albedo.z * om);
    f3 ks = mk3(0.04f * om + m.albedo.x * m.metal, 0.04f * om + m.albedo.y * m.metal, 0.04f * om + m.albedo.z * m.metal);
    float G = G1_GGX_Schlick(NoL, m.rough) * G1_GGX_Schlick(NoV, m.rough);   // G_Smith(NoV,NoL,r) = g1_l*g1_v
    float x1 = 1.f - VoH, x2 = x1 * x1, x = x2 * x2 * x1;                     // (1-VoH).pow(5)
    // fresnelSchlick(VoH,F0) = F0 + (1-F0)*x ; brdf_spec = D*G*F/4.0*NoL ; brdf_diff = kd/pi*NoL
    float dg = D * G;
    brdf.x = kd.x / kPi * NoL + dg * (ks.x + (1.f - ks.x) * x) / 4.0f * NoL;
    brdf.y = kd.y / kPi * NoL + dg * (ks.y + (1.f - ks.y) * x) / 4.0f * NoL;
    brdf.z = kd.z / kPi * NoL + dg * (ks.z + (1.f - ks.z) * x) / 4.0f * NoL;
}

// model/brdf.py:177-210 sample_brdf
__device__ __forceinline__ void sample_brdf1(float s1, float u0, float u1, f3 wo, f3 n, Mat m, f3& wi, float& pdf, f3& weight) {
    f3 t, b;
    normal_space(n, t, b);
    wi = (s1 > 0.5f) ? diffuse_sampler(u0, u1, n, t, b) : specular_sampler(u0, u1, m.rough, wo, n, t, b);
    f3 brdf;
    eval_brdf1(wi, wo, n, m, brdf, pdf);
    // torch.where(pdf>0, brdf/pdf, 0); NaN -> 0
    weight = mk3(0.f, 0.f, 0.f);
    if (pdf > 0.f) {
        weight = mk3(brdf.x / pdf, brdf.y / pdf, brdf.z / pdf);
        if (weight.x != weight.x) weight.x = 0.f;
        if (weight.y != weight.y) weight.y = 0.f;
        if (weight.z != weight.z) weight.z = 0.f;
    }
}

// model/emitter.py:224-255 sample_emitter
__device__ __forceinline__ void sample_emitter1(const EmitSampleDev& e, float s1, float u0, float u1, f3 pos, f3& wi, float& pdf, int64_t& tri) {
    const float v = fmaxf(s1, 1e-12f);
    int64_t lo = 0, hi = e.k;                     // torch.searchsorted(cdf, v): first i with cdf[i] >= v
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (e.cdf[mid] < v) lo = mid + 1; else hi = mid; }
    const int64_t ei = lo < e.k ? lo : e.k - 1;    // the reference would index out of range when v > cdf[-1]; clamp
    const float xi1 = sqrtf(u0);
    const float u = 1.f - xi1, vv = xi1 * u1, w = (1.f - u) - vv;
    const float* p = e.verts + ei * 9;
    f3 p1 = mk3((p[0] * u + p[3] * vv) + p[6] * w, (p[1] * u + p[4] * vv) + p[7] * w, (p[2] * u + p[5] * vv) + p[8] * w);
    wi = t_normalize(sub3(p1, pos));
    pdf = e.emitter_pdf / fmaxf(e.area[ei], 1e-12f);
    tri = e.ord2tri[ei];
}

// utils/path_tracing.py:338-340: wi = normalize(rays_d + dx_du*du + dy_dv*dv), du,dv = rand - 0.5
__global__ void pt_jitter_kernel(const float* __restrict__ rays_d, const float* __restrict__ dxdu, const float* __restrict__ dydv,
                                 const float* __restrict__ dudv /* (2,B,spp) */, int64_t B, int spp, float* __restrict__ wi) {
    const int64_t n = B * spp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / spp;
        const float du = dudv[i] - 0.5f, dv = dudv[n + i] - 0.5f;
        f3 d = ld3(rays_d + b * 3), dx = ld3(dxdu + b * 3), dy = ld3(dydv + b * 3);
        st3(wi + i * 3, t_normalize(mk3((d.x + dx.x * du) + dy.x * dv, (d.y + dx.y * du) + dy.y * dv, (d.z + dx.z * du) + dy.z * dv)));
    }
}

struct PtArgs {
    SceneDev sc; EmitDev em; SlfDev slf; EmitSampleDev es;
    int64_t N;
    const float *pos, *nrm, *wo, *albedo, *rough, *metal;   // (N,3),(N,3),(N,3),(N,3),(N),(N)
    const float *s1, *s2;                                    // (N),(N,2) uniforms
    const float *s1b, *s2b;                                  // pt_bounce_kernel: the BRDF stage's draws (s1 / s2 are the emitter-sampling stage's)
    // NEE outputs
    float* coef1; int32_t* e1;
    // BRDF-sample outputs
    float *wi_out, *brdf_pdf, *brdf_w, *pos_next, *nrm_next; int64_t* tri_next; uint8_t* valid_next_hit;
    // finish inputs/outputs
    const float *wi_in, *pdf_in, *w_in, *pos_n_in, *nrm_n_in, *rough_next; const int64_t* tri_n_in;
    float *coef2, *const2; int32_t* e2;
    // variant parameters: path_tracing_single (:320-407) clamps with 1e-6, trace_indirect (:409-502) with 1e-12 and no MIS clamp
    float g_eps, pdf_eps, mis_eps;   // mis_eps <= 0: no clamp_min on the NEE MIS denominator
    float trace_rough;               // eval_emitter's trace_roughness in the finish stage
    int lobe; float lobe_rough;      // brdf_trace: 0 = sample_brdf, 1 = sample_diffuse (weight 1), 2 = sample_specular(lobe_rough) -> weight (g0,g1,0)
};

__device__ __forceinline__ Mat load_mat(const PtArgs& a, int64_t i) {
    Mat m; m.albedo = ld3(a.albedo + i * 3); m.rough = a.rough[i]; m.metal = a.metal[i];
    return m;
}

// hit triangle of a leaf slot: vertices + original triangle index
__device__ __forceinline__ void hit_triangle(const SceneDev& sc, int slot, f3& p0, f3& p1, f3& p2, int& id) {
    const float4* r = sc.tris + (int64_t)slot * 4;
    const float4 X = r[0], Y = r[1], Z = r[2];      // component-major record (iris_trace.h)
    p0 = mk3(X.x, Y.x, Z.x); p1 = mk3(X.y, Y.y, Z.y); p2 = mk3(X.z, Y.z, Z.z);
    id = __float_as_int(X.w);
}

// utils/path_tracing.py:357-382: emitter sampling, visibility ray, geometry term, eval_brdf, power-2 MIS.
// term1 = coef1 * radiance[e1]   (e1 = -1 -> no contribution)
// everything after the visibility ray: (slot, u, v) = its closest hit (slot < 0: miss)
__device__ __forceinline__ void pt_nee_finish(const PtArgs& a, int64_t i, f3 x, f3 n, f3 wo, f3 wi, float emit_pdf, int64_t emit_tri, int slot, float u,
                                              float v) {
    const bool emit_valid = slot >= 0;
    int ord = -1;
    float G = 1.f;
    bool emit_vis = true;                                    // (~emit_valid) | (emit_triangle_idx == triangle_idx)
    if (emit_valid) {
        f3 p0, p1, p2; int id;
        hit_triangle(a.sc, slot, p0, p1, p2, id);
        Hit h; h.u = u; h.v = v; h.slot = slot; h.t = 0.f; h.id = id;
        const f3 ep = hit_position(h, p0, p1, p2);
        f3 en = t_normalize(hit_normal(p0, p1, p2));
        if (t_dot(en, mk3(-wi.x, -wi.y, -wi.z)) < 0.f) en = mk3(-en.x, -en.y, -en.z);
        emit_vis = emit_tri == (int64_t)id;
        ord = a.em.emit_ord[id];                             // eval_emitter(emit_position, wi, triangle_idx): Le = radiance[ord] if emitter
        const f3 dlt = sub3(ep, x);
        const float d2 = (dlt.x * dlt.x + dlt.y * dlt.y) + dlt.z * dlt.z;
        G = fabsf(t_dot(mk3(-wi.x, -wi.y, -wi.z), en)) / fmaxf(d2, a.g_eps);
    }
    f3 brdf; float brdf_pdf;
    eval_brdf1(wi, wo, n, load_mat(a, i), brdf, brdf_pdf);
    brdf_pdf = brdf_pdf * G;
    float w_mis = 0.f;
    if (emit_pdf > 0.f && !isinf(brdf_pdf)) {
        float den = emit_pdf * emit_pdf + brdf_pdf * brdf_pdf;
        if (a.mis_eps > 0.f) den = fmaxf(den, a.mis_eps);
        w_mis = emit_pdf * emit_pdf / den;
    }
    if (isinf(emit_pdf) || brdf_pdf == 0.f) w_mis = 1.f;
    // emit_weight = Le * emit_vis * G / clamp(emit_pdf,eps); L += emit_brdf * emit_weight * w_mis
    const float s = (emit_vis ? 1.f : 0.f);
    const float ew = G / fmaxf(emit_pdf, a.pdf_eps);
    // coefficient applied to radiance[ord]: ((1*vis)*G/pdf) then *brdf then *w_mis, in the reference's evaluation order
    st3(a.coef1 + i * 3, mk3(brdf.x * (s * ew) * w_mis, brdf.y * (s * ew) * w_mis, brdf.z * (s * ew) * w_mis));
    a.e1[i] = (emit_valid && ord >= 0) ? ord : -1;
}

#ifndef IRIS_JOINT_WAVES
#define IRIS_JOINT_WAVES 5      // min waves per SIMD the latency-mode instantiations are compiled for (unlimited: 104 VGPRs = 4 waves; 5: 96 VGPRs, no spills -- the NEE and the BRDF stage of a call run side by side on two streams: cfg 5 447-457 / 464-466 / 444-459 Mpaths/s at 4 / 5 / 6)
#endif
template <int LAYOUT, bool JOINT = false>
__global__ __launch_bounds__(kBlock, JOINT ? IRIS_JOINT_WAVES : 1) void pt_nee_kernel(PtArgs a) {
    __shared__ uint32_t s_stack[kStackLds * kBlock];
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < a.N; i += (int64_t)gridDim.x * kBlock) {
        const f3 x = ld3(a.pos + i * 3), n = ld3(a.nrm + i * 3), wo = ld3(a.wo + i * 3);
        f3 wi; float emit_pdf; int64_t emit_tri;
        sample_emitter1(a.es, a.s1[i], a.s2[i * 2], a.s2[i * 2 + 1], x, wi, emit_pdf, emit_tri);
        const f3 o = mk3(x.x + kRayEps * wi.x, x.y + kRayEps * wi.y, x.z + kRayEps * wi.z);
        Hit h = trace_bvh4<LAYOUT, false, kStackLds, false, JOINT>(a.sc, o, wi, s_stack + threadIdx.x);
        pt_nee_finish(a, i, x, n, wo, wi, emit_pdf, emit_tri, h.slot, h.u, h.v);
    }
}

// utils/path_tracing.py:384-392: BRDF sampling + next intersection
// direction, pdf and weight of ray i (lobe 0: sample_brdf; 1: sample_diffuse; 2: sample_specular at lobe_rough)
__device__ __forceinline__ void pt_sample_dir(const PtArgs& a, int64_t i, f3 wo, f3 n, f3& wi, float& pdf, f3& w, const float* s1 = nullptr, const float* s2 = nullptr) {
    if (!s2) { s1 = a.s1; s2 = a.s2; }
    if (a.lobe == 0) {
        sample_brdf1(s1[i], s2[i * 2], s2[i * 2 + 1], wo, n, load_mat(a, i), wi, pdf, w);
    } else {
        f3 t, b;
        normal_space(n, t, b);
        if (a.lobe == 1) {                                   // BaseBRDF.sample_diffuse (model/brdf.py:78-88)
            wi = diffuse_sampler(s2[i * 2], s2[i * 2 + 1], n, t, b);
            pdf = relu(t_dot(n, wi)) / kPi;
            w = mk3(1.f, 1.f, 1.f);
        } else {                                             // BaseBRDF.sample_specular (model/brdf.py:112-136)
            wi = specular_sampler(s2[i * 2], s2[i * 2 + 1], a.lobe_rough, wo, n, t, b);
            SpecW sw = specular_weights(wi, wo, n, a.lobe_rough, true);
            pdf = sw.pdf;
            w = mk3(sw.g0, sw.g1, 0.f);
        }
    }
}
// next-hit outputs of ray i from its closest hit (slot, u, v)
__device__ __forceinline__ void pt_next_hit(const PtArgs& a, int64_t i, f3 wi, int slot, float u, float v) {
    f3 pn = mk3(0.f, 0.f, 0.f), nn = mk3(0.f, 0.f, 0.f);
    int64_t tri = -1;
    if (slot >= 0) {
        f3 p0, p1, p2; int id;
        hit_triangle(a.sc, slot, p0, p1, p2, id);
        Hit h; h.u = u; h.v = v; h.slot = slot; h.t = 0.f; h.id = id;
        pn = hit_position(h, p0, p1, p2);
        nn = t_normalize(hit_normal(p0, p1, p2));
        if (t_dot(nn, mk3(-wi.x, -wi.y, -wi.z)) < 0.f) nn = mk3(-nn.x, -nn.y, -nn.z);
        tri = id;
    }
    st3(a.pos_next + i * 3, pn); st3(a.nrm_next + i * 3, nn); a.tri_next[i] = tri; a.valid_next_hit[i] = slot >= 0;
}

template <int LAYOUT, bool JOINT = false>
__global__ __launch_bounds__(kBlock, JOINT ? IRIS_JOINT_WAVES : 1) void pt_brdf_trace_kernel(PtArgs a) {
    __shared__ uint32_t s_stack[kStackLds * kBlock];
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < a.N; i += (int64_t)gridDim.x * kBlock) {
        const f3 x = ld3(a.pos + i * 3), n = ld3(a.nrm + i * 3), wo = ld3(a.wo + i * 3);
        f3 wi, w; float pdf;
        pt_sample_dir(a, i, wo, n, wi, pdf, w);
        const f3 o = mk3(x.x + kRayEps * wi.x, x.y + kRayEps * wi.y, x.z + kRayEps * wi.z);
        Hit h = trace_bvh4<LAYOUT, false, kStackLds, false, JOINT>(a.sc, o, wi, s_stack + threadIdx.x);
        st3(a.wi_out + i * 3, wi); a.brdf_pdf[i] = pdf; st3(a.brdf_w + i * 3, w);
        pt_next_hit(a, i, wi, h.slot, h.u, h.v);
    }
}

// ---- large batches (refine_shading's whole-image calls): the same two stages through the tile machinery of iris_tile.h --
// rays binned by direction per tile, persistent-lane traversal.  Nothing needs a workspace: the sampled direction is parked in an
// output array of the stage (NEE: coef1, BRDF stage: wi_out), the hit (u, v, leaf slot) in another (NEE: coef1 + e1, BRDF stage:
// pos_next) until the epilogue overwrites them with the final values.  Same per-ray arithmetic as the kernels above: same bits.
constexpr int kPtTileCap = 4096, kPtTileStack = 10;
#ifndef IRIS_PT_WAVES
#define IRIS_PT_WAVES 6
#endif

template <int LAYOUT, bool NEE>
__global__ __launch_bounds__(kBlock, IRIS_PT_WAVES) void pt_tiled_kernel(PtArgs a, int tile_rays) {
    __shared__ uint16_t s_sorted[kPtTileCap];
    __shared__ uint32_t s_stack[kPtTileStack * kBlock];
    __shared__ int s_chunk;
    const int tid = threadIdx.x;
    const int64_t n_tiles = (a.N + tile_rays - 1) / tile_rays;
    TraceStats ts;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        if (tid == 0) s_chunk = 0;
        (s_stack + kPtTileCap / 4)[tid] = 0;
        __syncthreads();
        const int64_t i0 = tile * tile_rays;
        const int nr = (int)min((int64_t)tile_rays, a.N - i0);
        float* dir = NEE ? a.coef1 : a.wi_out;       // where the direction is parked
        float* rec = NEE ? a.coef1 : a.pos_next;     // where (u, v[, slot]) is parked
        tile_sort_trace<LAYOUT, false, kPtTileCap, kPtTileStack, false>(
            a.sc, nr, s_sorted, s_stack, &s_chunk, nullptr, ts,
            [&](int r) -> uint32_t {
                const int64_t i = i0 + r;
                f3 wi;
                if (NEE) {
                    float emit_pdf; int64_t emit_tri;
                    sample_emitter1(a.es, a.s1[i], a.s2[i * 2], a.s2[i * 2 + 1], ld3(a.pos + i * 3), wi, emit_pdf, emit_tri);
                } else {
                    f3 w; float pdf;
                    pt_sample_dir(a, i, ld3(a.wo + i * 3), ld3(a.nrm + i * 3), wi, pdf, w);
                    a.brdf_pdf[i] = pdf; st3(a.brdf_w + i * 3, w);
                }
                st3(dir + i * 3, wi);
                return dir_bin(wi);
            },
            [&](int r, f3& o, f3& d) { const int64_t i = i0 + r; o = ld3(a.pos + i * 3); d = ld3(dir + i * 3); },
            [&](f3& o, f3& d) { o = mk3(o.x + kRayEps * d.x, o.y + kRayEps * d.y, o.z + kRayEps * d.z); },
            [&](int r, const Hit& h) {
                const int64_t i = i0 + r;
                if (NEE) { rec[i * 3] = h.u; rec[i * 3 + 1] = h.v; a.e1[i] = h.slot; }
                else st3(rec + i * 3, mk3(h.u, h.v, __int_as_float(h.slot)));
            });
        // epilogue: final outputs from the parked hit
        for (int r = tid; r < nr; r += kBlock) {
            const int64_t i = i0 + r;
            if (NEE) {
                const f3 x = ld3(a.pos + i * 3);
                f3 wi; float emit_pdf; int64_t emit_tri;
                sample_emitter1(a.es, a.s1[i], a.s2[i * 2], a.s2[i * 2 + 1], x, wi, emit_pdf, emit_tri);   // (the direction slot now holds u, v)
                pt_nee_finish(a, i, x, ld3(a.nrm + i * 3), ld3(a.wo + i * 3), wi, emit_pdf, emit_tri, a.e1[i], rec[i * 3], rec[i * 3 + 1]);
            } else {
                const f3 h = ld3(rec + i * 3);
                pt_next_hit(a, i, ld3(a.wi_out + i * 3), __float_as_int(h.z), h.x, h.y);
            }
        }
    }
}

// Both ray kinds of a bounce behind ONE launch (round 6; trace_indirect, utils/path_tracing.py:434-471): the emitter-sampling stage's visibility ray and the BRDF
// stage's ray of a path leave from the same point, and a whole-image bounce of refine_shading's reference batch (1.3 M paths) cut into two launches leaves each
// with 512-ray tiles.  A tile here is np PATHS = 2 np rays -- ray r < np: the emitter-sampling ray of path i0 + r, ray r >= np: the BRDF ray of path i0 + r - np --
// sorted by direction TOGETHER and traced by the same persistent lanes; the parked values and the epilogues are those of pt_tiled_kernel<NEE> and <!NEE>: same bits.
template <int LAYOUT>
__global__ __launch_bounds__(kBlock, IRIS_PT_WAVES) void pt_bounce_kernel(PtArgs a, int tile_paths) {
    __shared__ uint16_t s_sorted[kPtTileCap];
    __shared__ uint32_t s_stack[kPtTileStack * kBlock];
    __shared__ int s_chunk;
    const int tid = threadIdx.x;
    const int64_t n_tiles = (a.N + tile_paths - 1) / tile_paths;
    TraceStats ts;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        if (tid == 0) s_chunk = 0;
        (s_stack + kPtTileCap / 4)[tid] = 0;
        __syncthreads();
        const int64_t i0 = tile * tile_paths;
        const int np = (int)min((int64_t)tile_paths, a.N - i0);
        tile_sort_trace<LAYOUT, false, kPtTileCap, kPtTileStack, false>(
            a.sc, 2 * np, s_sorted, s_stack, &s_chunk, nullptr, ts,
            [&](int r) -> uint32_t {
                const bool brdf = r >= np;
                const int64_t i = i0 + (brdf ? r - np : r);
                f3 wi;
                if (!brdf) {
                    float emit_pdf; int64_t emit_tri;
                    sample_emitter1(a.es, a.s1[i], a.s2[i * 2], a.s2[i * 2 + 1], ld3(a.pos + i * 3), wi, emit_pdf, emit_tri);
                    st3(a.coef1 + i * 3, wi);
                } else {
                    f3 w; float pdf;
                    pt_sample_dir(a, i, ld3(a.wo + i * 3), ld3(a.nrm + i * 3), wi, pdf, w, a.s1b, a.s2b);
                    a.brdf_pdf[i] = pdf; st3(a.brdf_w + i * 3, w);
                    st3(a.wi_out + i * 3, wi);
                }
                return dir_bin(wi);
            },
            [&](int r, f3& o, f3& d) { const bool brdf = r >= np; const int64_t i = i0 + (brdf ? r - np : r); o = ld3(a.pos + i * 3); d = ld3((brdf ? a.wi_out : a.coef1) + i * 3); },
            [&](f3& o, f3& d) { o = mk3(o.x + kRayEps * d.x, o.y + kRayEps * d.y, o.z + kRayEps * d.z); },
            [&](int r, const Hit& h) {
                const bool brdf = r >= np;
                const int64_t i = i0 + (brdf ? r - np : r);
                if (!brdf) { a.coef1[i * 3] = h.u; a.coef1[i * 3 + 1] = h.v; a.e1[i] = h.slot; }
                else st3(a.pos_next + i * 3, mk3(h.u, h.v, __int_as_float(h.slot)));
            });
        for (int r = tid; r < 2 * np; r += kBlock) {
            const bool brdf = r >= np;
            const int64_t i = i0 + (brdf ? r - np : r);
            if (!brdf) {
                const f3 x = ld3(a.pos + i * 3);
                f3 wi; float emit_pdf; int64_t emit_tri;
                sample_emitter1(a.es, a.s1[i], a.s2[i * 2], a.s2[i * 2 + 1], x, wi, emit_pdf, emit_tri);   // (the direction slot now holds u, v)
                pt_nee_finish(a, i, x, ld3(a.nrm + i * 3), ld3(a.wo + i * 3), wi, emit_pdf, emit_tri, a.e1[i], a.coef1[i * 3], a.coef1[i * 3 + 1]);
            } else {
                const f3 h = ld3(a.pos_next + i * 3);
                pt_next_hit(a, i, ld3(a.wi_out + i * 3), __float_as_int(h.z), h.x, h.y);
            }
        }
    }
}

// utils/path_tracing.py:394-404: eval_emitter at the BRDF-sampled hit, geometry term, MIS.
// term2 = coef2 * radiance[e2] + const2   (const2 = coef2 * SLF radiance)
__global__ void pt_brdf_finish_kernel(PtArgs a) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.N; i += (int64_t)gridDim.x * blockDim.x) {
        const f3 x = ld3(a.pos + i * 3), pn = ld3(a.pos_n_in + i * 3), nn = ld3(a.nrm_n_in + i * 3), wi = ld3(a.wi_in + i * 3);
        const int64_t tri = a.tri_n_in[i];
        // eval_emitter(position_next, wi, triangle_idx, mat_next['roughness'], trace_roughness=0.0), radiance factored out
        const bool vis = tri != -1;
        int ord = -1;
        if (vis) ord = a.em.emit_ord[tri];
        const bool is_area = ord >= 0;
        float emit_pdf = 0.f;
        if (is_area) emit_pdf = a.em.emitter_pdf / fmaxf(a.em.area[ord], 1e-12f);
        bool valid_next = (!is_area) && vis;
        f3 slf = mk3(0.f, 0.f, 0.f);
        if ((!is_area) && vis && (!a.rough_next || a.rough_next[i] > a.trace_rough)) {       // (rough_next NULL: the caller KNOWS that every roughness exceeds trace_roughness)
            slf = slf_forward(a.slf, pn);
            if ((slf.x + slf.y) + slf.z > 0.f) valid_next = false;
        }
        const f3 dlt = sub3(x, pn);
        const float d2 = (dlt.x * dlt.x + dlt.y * dlt.y) + dlt.z * dlt.z;
        float G = fabsf(t_dot(mk3(-nn.x, -nn.y, -nn.z), wi)) / fmaxf(d2, a.g_eps);
        if (!valid_next) G = 1.f;                                 // torch.where(valid_next, G, 1)
        if (a.valid_next_hit) a.valid_next_hit[i] = valid_next ? 1 : 0;
        const float brdf_pdf = a.pdf_in[i] * G;
        float w_mis = 0.f;
        if (brdf_pdf > 0.f && !isinf(emit_pdf)) w_mis = brdf_pdf * brdf_pdf / (emit_pdf * emit_pdf + brdf_pdf * brdf_pdf);
        if (isinf(brdf_pdf) || emit_pdf == 0.f) w_mis = 1.f;
        const f3 w = ld3(a.w_in + i * 3);
        // L += brdf_weight * Le * w_mis with Le = radiance[ord] (area) + slf (diffuse cache)
        st3(a.coef2 + i * 3, mk3(w.x * w_mis, w.y * w_mis, w.z * w_mis));
        st3(a.const2 + i * 3, mk3(w.x * slf.x * w_mis, w.y * slf.y * w_mis, w.z * slf.z * w_mis));
        a.e2[i] = is_area ? ord : -1;
    }
}

// L[b] = mean_s( radiance[e0] + [active] (coef1*radiance[e1] + (coef2*radiance[e2] + const2)) )   (utils/path_tracing.py:344,382,404,406)
// path_of: (B*spp) int32 index into the compacted arrays or -1.
// One LANE per (pixel, sample): the terms of a pixel are gathered in parallel -- five dependent loads each -- and then added in the order s = 0, 1, ... by every lane of the
// pixel's group through lane reads, i.e. the SAME sequential float sum as a one-thread-per-pixel loop (the bits of round 1-4's kernel and of the oracle), without its spp
// dependent round trips per thread (8192 threads x 32 trips: 56 us of a 0.56 ms cfg-5 call; now ~8 us).  lpp = lanes per pixel = min(64, next power of two >= spp).
__global__ __launch_bounds__(256) void pt_accumulate_fwd_kernel(const float* __restrict__ radiance, const int32_t* __restrict__ e0, const int32_t* __restrict__ path_of,
                                         const int32_t* __restrict__ e1, const float* __restrict__ coef1, const int32_t* __restrict__ e2,
                                         const float* __restrict__ coef2, const float* __restrict__ const2, int64_t B, int spp, int lpp,
                                         float* __restrict__ L) {
    const int lane = threadIdx.x & 63, sub = lane / lpp, sl = lane - sub * lpp, ppw = 64 / lpp;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t n_groups = (B + ppw - 1) / ppw;
    const float inv = 1.0f / (float)spp;
    for (int64_t g = wave; g < n_groups; g += n_waves) {
        const int64_t b = g * ppw + sub;
        float ax = 0.f, ay = 0.f, az = 0.f;
        for (int s0 = 0; s0 < spp; s0 += lpp) {                      // (spp > 64: rounds of 64 samples, still in order)
            const int sidx = s0 + sl;
            f3 l = mk3(0.f, 0.f, 0.f);
            if (b < B && sidx < spp) {
                const int64_t i = b * spp + sidx;
                if (e0[i] >= 0) l = ld3(radiance + (int64_t)e0[i] * 3);
                const int j = path_of[i];
                if (j >= 0) {
                    if (e1[j] >= 0) { f3 r = ld3(radiance + (int64_t)e1[j] * 3), c = ld3(coef1 + (int64_t)j * 3); l.x += c.x * r.x; l.y += c.y * r.y; l.z += c.z * r.z; }
                    f3 t2 = ld3(const2 + (int64_t)j * 3);
                    if (e2[j] >= 0) { f3 r = ld3(radiance + (int64_t)e2[j] * 3), c = ld3(coef2 + (int64_t)j * 3); t2.x += c.x * r.x; t2.y += c.y * r.y; t2.z += c.z * r.z; }
                    l.x += t2.x; l.y += t2.y; l.z += t2.z;
                }
            }
            const int n = min(lpp, spp - s0);                        // (wave-uniform)
            for (int k = 0; k < n; ++k) {                            // the sequential sum, by every lane of the group (lane reads within the group)
                const int src = sub * lpp + k;
                ax += __shfl(l.x, src); ay += __shfl(l.y, src); az += __shfl(l.z, src);
            }
        }
        if (b < B && sl == 0) st3(L + b * 3, mk3(ax * inv, ay * inv, az * inv));
    }
}
// d radiance[e] += gL[b]/spp * coef   (scatter-add; few thousand rows, contention is irrelevant at 2.6e5 paths)
__global__ void pt_accumulate_bwd_kernel(const float* __restrict__ gL, const int32_t* __restrict__ e0, const int32_t* __restrict__ path_of,
                                         const int32_t* __restrict__ e1, const float* __restrict__ coef1, const int32_t* __restrict__ e2,
                                         const float* __restrict__ coef2, int64_t B, int spp, float* __restrict__ g_radiance) {
    const int64_t n = B * spp;
    const float inv = 1.0f / (float)spp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / spp;
        const f3 g = mk3(gL[b * 3] * inv, gL[b * 3 + 1] * inv, gL[b * 3 + 2] * inv);
        if (e0[i] >= 0) { float* q = g_radiance + (int64_t)e0[i] * 3; atomicAdd(q, g.x); atomicAdd(q + 1, g.y); atomicAdd(q + 2, g.z); }
        const int j = path_of[i];
        if (j >= 0) {
            if (e1[j] >= 0) { float* q = g_radiance + (int64_t)e1[j] * 3; f3 c = ld3(coef1 + (int64_t)j * 3); atomicAdd(q, g.x * c.x); atomicAdd(q + 1, g.y * c.y); atomicAdd(q + 2, g.z * c.z); }
            if (e2[j] >= 0) { float* q = g_radiance + (int64_t)e2[j] * 3; f3 c = ld3(coef2 + (int64_t)j * 3); atomicAdd(q, g.x * c.x); atomicAdd(q + 1, g.y * c.y); atomicAdd(q + 2, g.z * c.z); }
        }
    }
}

// L[rows[i]] += throughput[i] * (coef[i] * radiance[e[i]] + cst[i]), NaN -> 0 (trace_indirect's `dL[dL.isnan()] = 0`, :454-456,:484-486);
// then optionally throughput[i] *= weight[i] (:462).  rows / throughput / cst / weight may be NULL.
__global__ void pt_apply_kernel(float* __restrict__ Lacc, const int32_t* __restrict__ rows, float* __restrict__ throughput,
                                const float* __restrict__ radiance, const int32_t* __restrict__ e, const float* __restrict__ coef,
                                const float* __restrict__ cst, const float* __restrict__ weight, int64_t N, int nan_to_zero) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        f3 v = cst ? ld3(cst + i * 3) : mk3(0.f, 0.f, 0.f);
        if (e && e[i] >= 0) { f3 r = ld3(radiance + (int64_t)e[i] * 3), c = ld3(coef + i * 3); v = mk3(v.x + c.x * r.x, v.y + c.y * r.y, v.z + c.z * r.z); }
        if (throughput) { f3 t = ld3(throughput + i * 3); v = mk3(t.x * v.x, t.y * v.y, t.z * v.z); }
        if (nan_to_zero) { if (v.x != v.x) v.x = 0.f; if (v.y != v.y) v.y = 0.f; if (v.z != v.z) v.z = 0.f; }
        float* q = Lacc + (int64_t)(rows ? rows[i] : i) * 3;
        q[0] += v.x; q[1] += v.y; q[2] += v.z;                     // rows are unique: no atomics needed
        if (throughput && weight) { f3 t = ld3(throughput + i * 3), w = ld3(weight + i * 3); st3(throughput + i * 3, mk3(t.x * w.x, t.y * w.y, t.z * w.z)); }
    }
}

// trace_indirect's end of a bounce (utils/path_tracing.py:488-501: `active_next[...] = valid_next; position = position[valid_next]; ...`): the rows with keep[i] != 0 move to
// the front of the output arrays IN ORDER (boolean indexing keeps the order; the recorded draws of the parity fixtures are consumed in it).  Two launches, no host
// round trip, no index tensor: (1) every workgroup counts the kept rows of its kCompactItems consecutive rows; (2) every workgroup sums the counts in front of it,
// scans its own rows and moves them.  Up to kCompactMax arrays of each kind (3 floats / 1 float / 1 int32 per row); negate3 bit k: dst3[k] = -src3[k] (wo = -wi).
constexpr int kCompactMax = 6, kCompactRounds = 8, kCompactItems = 256 * kCompactRounds;   // rows per workgroup: every wave takes 8 runs of 64 CONSECUTIVE rows (lane = row: coalesced)
struct CompactArgs {
    const uint8_t* keep; int64_t N;
    int n3, n1, ni; uint32_t negate3;
    const float* src3[kCompactMax]; float* dst3[kCompactMax];
    const float* src1[kCompactMax]; float* dst1[kCompactMax];
    const int32_t* srci[kCompactMax]; int32_t* dsti[kCompactMax];
    int32_t* block_counts; int32_t* count;
};
// rows kept among the 64 x kCompactRounds rows of this wave (lane = row inside a run)
__device__ __forceinline__ int compact_wave_count(const CompactArgs& a, int64_t w0, int lane) {
    int c = 0;
#pragma unroll
    for (int k = 0; k < kCompactRounds; ++k) {
        const int64_t i = w0 + (int64_t)k * 64 + lane;
        c += __popcll(__ballot(i < a.N && a.keep[i]));
    }
    return c;      // wave-uniform
}
__global__ __launch_bounds__(256) void pt_compact_count_kernel(CompactArgs a) {
    __shared__ int s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = compact_wave_count(a, (int64_t)blockIdx.x * kCompactItems + (int64_t)wave * (64 * kCompactRounds), lane);
    if (lane == 0) s_w[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) a.block_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void pt_compact_move_kernel(CompactArgs a) {
    __shared__ int s_w[4], s_c[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // rows kept by the workgroups in front of this one
    int before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += a.block_counts[b];
    for (int m = 1; m < 64; m <<= 1) before += __shfl_xor(before, m);
    const int64_t w0 = (int64_t)blockIdx.x * kCompactItems + (int64_t)wave * (64 * kCompactRounds);
    const int mine = compact_wave_count(a, w0, lane);
    if (lane == 0) { s_w[wave] = before; s_c[wave] = mine; }
    __syncthreads();
    int64_t off = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    for (int w = 0; w < wave; ++w) off += s_c[w];
    if (blockIdx.x == gridDim.x - 1 && wave == 3 && lane == 0) *a.count = (int32_t)(off + mine);
    for (int k = 0; k < kCompactRounds; ++k) {
        const int64_t i = w0 + (int64_t)k * 64 + lane;
        const bool kp = i < a.N && a.keep[i];
        const unsigned long long m = __ballot(kp);
        if (kp) {
            const int64_t o = off + __popcll(m & ((1ull << lane) - 1ull));
            for (int j = 0; j < a.n3; ++j) {
                f3 v = ld3(a.src3[j] + i * 3);
                if ((a.negate3 >> j) & 1u) v = mk3(-v.x, -v.y, -v.z);
                st3(a.dst3[j] + o * 3, v);
            }
            for (int j = 0; j < a.n1; ++j) a.dst1[j][o] = a.src1[j][i];
            for (int j = 0; j < a.ni; ++j) a.dsti[j][o] = a.srci[j][i];
        }
        off += __popcll(m);
    }
}

// unfused call-surface kernels
__global__ void sample_emitter_kernel(EmitSampleDev e, const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ pos,
                                      int64_t N, float* __restrict__ wi, float* __restrict__ pdf, int64_t* __restrict__ tri) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        f3 w; float p; int64_t t;
        sample_emitter1(e, s1[i], s2[i * 2], s2[i * 2 + 1], ld3(pos + i * 3), w, p, t);
        st3(wi + i * 3, w); pdf[i] = p; tri[i] = t;
    }
}
__global__ void eval_brdf_kernel(const float* __restrict__ wi, const float* __restrict__ wo, const float* __restrict__ nrm, const float* __restrict__ albedo,
                                 const float* __restrict__ rough, const float* __restrict__ metal, int64_t N, float* __restrict__ brdf,
                                 float* __restrict__ pdf) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        Mat m; m.albedo = ld3(albedo + i * 3); m.rough = rough[i]; m.metal = metal[i];
        f3 b; float p;
        eval_brdf1(ld3(wi + i * 3), ld3(wo + i * 3), ld3(nrm + i * 3), m, b, p);
        st3(brdf + i * 3, b); pdf[i] = p;
    }
}
__global__ void sample_brdf_kernel(const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ wo, const float* __restrict__ nrm,
                                   const float* __restrict__ albedo, const float* __restrict__ rough, const float* __restrict__ metal, int64_t N,
                                   float* __restrict__ wi, float* __restrict__ pdf, float* __restrict__ weight) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        Mat m; m.albedo = ld3(albedo + i * 3); m.rough = rough[i]; m.metal = metal[i];
        f3 w, bw; float p;
        sample_brdf1(s1[i], s2[i * 2], s2[i * 2 + 1], ld3(wo + i * 3), ld3(nrm + i * 3), m, w, p, bw);
        st3(wi + i * 3, w); pdf[i] = p; st3(weight + i * 3, bw);
    }
}

}  // namespace iris
