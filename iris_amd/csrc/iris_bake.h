// Fused bake kernels (bake_shading.py:108-123 diffuse, :168-188 specular): uniforms -> BRDF sample -> secondary-ray
// traversal -> SLF / emitter lookup -> weights -> mean over spp.
//
//   bake_kernel       pixel-per-wave: one pixel per wave, lanes = samples (maximally incoherent rays inside a wave).  The simple
//                     reference implementation of the fixed reduction order; used for spp > 5120 and as the A/B baseline.
//   bake_tile_kernel  one lobe per launch: a persistent workgroup owns a TILE of consecutive pixels (~5120 rays), samples and bins
//                     the tile's rays by direction (octant-major bins, iris_tile.h) with an LDS counting sort, traces them in
//                     sorted order with persistent lanes (trace_stream), then shades and reduces each pixel's samples in the SAME
//                     fixed order as bake_kernel -> bit-identical outputs (tile_body below).
//   bake_view_kernel  all lobes of a view behind one persistent launch and one tile queue (same tile_body, same bits).
#pragma once
#include "iris_tile.h"

namespace iris {

struct BakeArgs {
    SceneDev sc; EmitDev em; SlfDev slf;
    const float* pos; const float* nrm; const float* wo;
    const float* u2; const int32_t* pix_id;
    int64_t P; int spp; uint64_t seed; uint32_t stream_id; float rough;
    float* out0; float* out1; int64_t* tri_next;
    int64_t* src_next;          // diagnostics (iris_hip_debug.h): per sample, the radiance-table row that was read (eval_emitter1's src)
    unsigned long long* stats;  // instrumented launches only: 20 slots {rays, node visits, tri tests, wave node iters, wave leaf iters, rays with
                                // stack > 8 / 12 / 16, (v1) tail sum, node visits / wave node iters while draining, [11..14] node visits with index < 21 / 85 / 341 / 1365, [15..17] wave node iterations with >= 32 lanes at one node / those lanes / iterations with ALL lanes at one node, [18] iterations with checked pushes, [19] iterations executed through the scalar path}
    // tile kernels only
    uint32_t* stack_ovf;        // gridDim.x * (kStackCapacity - LDS depth) * 256 dwords: traversal-stack entries beyond the LDS part
    float4* scratch;            // gridDim.x * kTileRays * (SPEC ? 2 : 1) float4: per-ray slots (sampled direction -> hit; GGX weights)
#if IRIS_PARK
    iris_u4v* park;             // gridDim.x * 4 waves * kParkCap records of 80 B (straggler parking, iris_trace.h)
#endif
    unsigned int* tile_counter; // 8 counters (one per XCD, claim_tile), zeroed before the launch
    int tile_px;                // pixels per tile (tile_px * spp <= kTileRays)
};

#ifndef IRIS_TILE_RAYS
#define IRIS_TILE_RAYS 5120
#endif
#ifndef IRIS_EXP_NOSAMPLE
#define IRIS_EXP_NOSAMPLE 0
#endif
#ifndef IRIS_EXP_NOSHADE
#define IRIS_EXP_NOSHADE 0
#endif
constexpr int kTileRays = IRIS_TILE_RAYS;   // capacity of the LDS ray list (10 KiB) = largest spp of the tile kernels; the host aims at tiles of this size

// lanes-per-pixel / pixels-per-wave geometry of the per-pixel reduction (shared by all bake kernels so that the sums match)
__device__ __forceinline__ void reduce_geometry(int spp, int& lpp, int& ppw, int& rounds) {
    if (spp >= 64) { lpp = 64; ppw = 1; rounds = (spp + 63) >> 6; }
    else if ((spp & (spp - 1)) == 0) { lpp = spp; ppw = 64 / spp; rounds = 1; }
    else { lpp = 64; ppw = 1; rounds = 1; }
}

struct RayOut { float r0, g0, b0, r1, g1, b1; };

// One (pixel, sample) in two halves.  sample_lobe: uniforms -> incident direction (+ the GGX weights of the specular lobe);
// trace_shade: trace it, evaluate the emitter / SLF at the hit, weight.  The tile kernels run the first half in their binning
// pass and park (wi, g0, g1) in the ray's result slot, so that the traversal pass carries only the origin and those five floats.
template <bool SPEC>
__device__ __forceinline__ void sample_dir(const BakeArgs& a, float u0, float u1, f3 n, f3 w, f3 t, f3 b, f3& wi, float& g0, float& g1) {
    g0 = 1.f; g1 = 0.f;
    if (SPEC) {
        wi = specular_sampler(u0, u1, a.rough, w, n, t, b);
        SpecW sw = specular_weights(wi, w, n, a.rough, false);
        g0 = sw.g0; g1 = sw.g1;
    } else {
        wi = diffuse_sampler(u0, u1, n, t, b);
    }
}
template <bool SPEC>
__device__ __forceinline__ void sample_lobe(const BakeArgs& a, int64_t p, int s, f3 n, f3 w, f3 t, f3 b, uint64_t base, f3& wi, float& g0, float& g1) {
    float u0, u1;
    if (a.u2) { const float* up = a.u2 + (p * a.spp + s) * 2; u0 = up[0]; u1 = up[1]; }
    else {
        // (the ten round keys are wave-uniform: left to itself hipcc hoists them out of the sampling loop into 20 scalar registers it does not have and
        //  reloads them from vector-register lanes every iteration; behind this barrier they are re-derived by scalar adds, which cost no vector issue)
        uint32_t s_lo = (uint32_t)a.seed, s_hi = (uint32_t)(a.seed >> 32);
        asm volatile("" : "+s"(s_lo), "+s"(s_hi));
        philox_u2(((uint64_t)s_hi << 32) | s_lo, base + (uint64_t)s, a.stream_id, u0, u1);
    }
    sample_dir<SPEC>(a, u0, u1, n, w, t, b, wi, g0, g1);
}

template <bool SPEC, bool COUNT, int LAYOUT, int LDS_DEPTH = kStackLds, bool GLOBAL_OVF = false>
__device__ __forceinline__ RayOut trace_shade(const BakeArgs& a, int64_t p, int s, f3 x, f3 wi, float g0, float g1, uint32_t* lds_stack,
                                              TraceStats* ts, uint32_t& n_rays, uint32_t* ovf = nullptr) {
    // position + RayEpsilon*wi  (bake_shading.py:117, :180)
    f3 o = mk3(x.x + kRayEps * wi.x, x.y + kRayEps * wi.y, x.z + kRayEps * wi.z);
    const uint32_t steps0 = COUNT ? ts->nodes + ts->tris : 0;
    Hit h = trace_bvh4<LAYOUT, COUNT, LDS_DEPTH, GLOBAL_OVF>(a.sc, o, wi, lds_stack, ts, ovf);
    if (COUNT) {
        n_rays++;
        // tail statistic: the wave lasts as long as its longest ray -> sum over waves of 64 * max(steps per lane)
        uint32_t mine = ts->nodes + ts->tris - steps0, mx = mine;
        for (int m = 1; m < 64; m <<= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, m));
        if (first_active_lane()) ts->max_steps64 += (unsigned long long)mx * 64ull;
    }
    f3 pn = mk3(0.f, 0.f, 0.f);
    int64_t tri = -1;
    if (h.slot >= 0) {
        f3 p0, p1, p2;
        hit_vertices(a.sc, h, p0, p1, p2);
        pn = hit_position(h, p0, p1, p2);
        tri = h.id;
    }
    if (a.tri_next) a.tri_next[p * a.spp + s] = tri;
    // eval_emitter(p_next, wi, tri_next, ones, trace_roughness=0.0)  (bake_shading.py:121-122, :184-185)
    float epdf; bool vn; int src;
    f3 Le = eval_emitter1(a.em, a.slf, pn, tri, true, 1.0f, 0.0f, epdf, vn, src);
    if (a.src_next) a.src_next[p * a.spp + s] = src;
    RayOut r;
    if (SPEC) { r.r0 = Le.x * g0; r.g0 = Le.y * g0; r.b0 = Le.z * g0; r.r1 = Le.x * g1; r.g1 = Le.y * g1; r.b1 = Le.z * g1; }
    else { r.r0 = Le.x; r.g0 = Le.y; r.b0 = Le.z; r.r1 = r.g1 = r.b1 = 0.f; }
    return r;
}

template <bool SPEC, bool COUNT, int LAYOUT, int LDS_DEPTH = kStackLds, bool GLOBAL_OVF = false>
__device__ __forceinline__ RayOut shade_sample(const BakeArgs& a, int64_t p, int s, f3 x, f3 n, f3 w, f3 t, f3 b, uint64_t base,
                                               uint32_t* lds_stack, TraceStats* ts, uint32_t& n_rays, uint32_t* ovf = nullptr) {
    f3 wi; float g0, g1;
    sample_lobe<SPEC>(a, p, s, n, w, t, b, base, wi, g0, g1);
    return trace_shade<SPEC, COUNT, LAYOUT, LDS_DEPTH, GLOBAL_OVF>(a, p, s, x, wi, g0, g1, lds_stack, ts, n_rays, ovf);
}

template <bool COUNT>
__device__ __forceinline__ void flush_stats(const BakeArgs& a, const TraceStats& ts, uint32_t n_rays) {
    if (COUNT) {
        uint32_t v[8] = {n_rays, ts.nodes, ts.tris, ts.node_iters, ts.leaf_iters, ts.sp_gt8, ts.sp_gt12, ts.sp_gt16};
        for (int k = 0; k < 8; ++k) {
            uint32_t x = v[k];
            for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m);
            if ((threadIdx.x & 63) == 0) atomicAdd(a.stats + k, (unsigned long long)x);
        }
        uint32_t w[2] = {ts.drain_nodes, ts.drain_node_iters};
        for (int k = 0; k < 2; ++k) {
            uint32_t x = w[k];
            for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m);
            if ((threadIdx.x & 63) == 0) atomicAdd(a.stats + 9 + k, (unsigned long long)x);
        }
        uint32_t tp[4] = {ts.top21, ts.top85, ts.top341, ts.top1365};
        for (int k = 0; k < 4; ++k) {
            uint32_t x = tp[k];
            for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m);
            if ((threadIdx.x & 63) == 0) atomicAdd(a.stats + 11 + k, (unsigned long long)x);
        }
        uint32_t sh[5] = {ts.shared_iters, ts.shared_lanes, ts.shared_all_iters, ts.slow_push_iters, ts.shared_path_iters};
        for (int k = 0; k < 5; ++k) {
            uint32_t x = sh[k];
            for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m);
            if ((threadIdx.x & 63) == 0) atomicAdd(a.stats + 15 + k, (unsigned long long)x);
        }
        unsigned long long y = ts.max_steps64;
        for (int m = 1; m < 64; m <<= 1) y += __shfl_xor(y, m);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.stats + 8, y);
    }
}

// ------------------------------------------------------------------------------------------------------- pixel-per-wave kernel
template <bool SPEC, bool COUNT, int LAYOUT>
__global__ __launch_bounds__(kBlock) void bake_kernel(BakeArgs a) {
    __shared__ uint32_t s_stack[kStackLds * kBlock];
    const int lane = threadIdx.x & 63;
    const int spp = a.spp;
    int lpp, ppw, rounds;
    reduce_geometry(spp, lpp, ppw, rounds);
    const int sub = lane / lpp, sl = lane - sub * lpp;
    const int64_t n_groups = (a.P + ppw - 1) / ppw;
    const int64_t wave0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
    const float inv_spp = 1.0f / (float)spp;
    TraceStats ts;
    uint32_t n_rays = 0;

    for (int64_t g = wave0; g < n_groups; g += n_waves) {
        const int64_t p = g * ppw + sub;
        const bool pvalid = p < a.P;
        f3 x = mk3(0.f, 0.f, 0.f), n = mk3(0.f, 0.f, 1.f), w = mk3(0.f, 0.f, 1.f), t, b;
        uint64_t base = 0;
        if (pvalid) {
            x = ld3(a.pos + p * 3); n = ld3(a.nrm + p * 3);
            if (SPEC) w = ld3(a.wo + p * 3);
            base = (uint64_t)(a.pix_id ? (int64_t)a.pix_id[p] : p) * (uint64_t)spp;
        }
        normal_space(n, t, b);
        float a0x = 0.f, a0y = 0.f, a0z = 0.f, a1x = 0.f, a1y = 0.f, a1z = 0.f;
        for (int r = 0; r < rounds; ++r) {
            const int s = r * 64 + sl;
            if (pvalid && s < spp) {
                RayOut o = shade_sample<SPEC, COUNT, LAYOUT>(a, p, s, x, n, w, t, b, base, s_stack + threadIdx.x, &ts, n_rays);
                a0x += o.r0; a0y += o.g0; a0z += o.b0;
                if (SPEC) { a1x += o.r1; a1y += o.g1; a1z += o.b1; }
            }
        }
        // .reshape(b,spp,3).mean(1): fixed butterfly over the lpp lanes of the pixel
        for (int m = 1; m < lpp; m <<= 1) {
            a0x += __shfl_xor(a0x, m); a0y += __shfl_xor(a0y, m); a0z += __shfl_xor(a0z, m);
            if (SPEC) { a1x += __shfl_xor(a1x, m); a1y += __shfl_xor(a1y, m); a1z += __shfl_xor(a1z, m); }
        }
        if (pvalid && sl == 0) {
            st3(a.out0 + p * 3, mk3(a0x * inv_spp, a0y * inv_spp, a0z * inv_spp));
            if (SPEC) st3(a.out1 + p * 3, mk3(a1x * inv_spp, a1y * inv_spp, a1z * inv_spp));
        }
    }
    flush_stats<COUNT>(a, ts, n_rays);
}

// ------------------------------------------------------------------------------------------------------- tile kernels
#ifndef IRIS_TILE_WAVES          // resident waves per SIMD the tile kernels are compiled for (= workgroups per CU): 7 x 20 488 B of LDS, 72 VGPRs.
#define IRIS_TILE_WAVES 7        // Measured: 6 waves (80 VGPRs) 7.11, 7 waves 7.24, 8 waves (64 VGPRs, 9-entry stacks) 7.17 Grays/s  (LDS now 7 x 22 536 B: 12-entry stacks)
#endif
#ifndef IRIS_RESOLVE_AT_RETIRE       // (A/B: 0 = a retiring ray parks (u, v, leaf slot) and the shading pass reads the triangle's record)
#define IRIS_RESOLVE_AT_RETIRE 1
#endif
#ifndef IRIS_TILE_STACK          // per-lane LDS stack entries of the tile kernels; deeper entries go to the workgroup's slab in the workspace
#define IRIS_TILE_STACK 12       // (a.stack_ovf), NOT to private scratch: without a scratch-resident stack array the kernel fits 6 (7) waves/SIMD (measured +6.5 %)
#endif

// One tile (<= kTileRays rays = the np <= tile_px consecutive valid pixels from p0 on, x spp) of one lobe, by one 256-thread workgroup.
//   LDS: s_sorted (10 KiB ray list) + s_stack (TILE_STACK KiB traversal stacks; doubles as the sort's key / histogram storage, the
//   two uses are separated by workgroup barriers) + *s_chunk (cursor into the sorted list).
//   res: the workgroup's slab of per-ray slots in the workspace: float4 res[kTileRays], then (specular) float2 res_g[kTileRays].
//   Slot life: phase A parks the sampled direction (res = wi) and the GGX weights (res_g = g1, g0); phase C replaces res by the
//   hit; phase D shades and sums.  The hit is parked RESOLVED when the records are the fused ones (iris_hip.hip fused_tris) and the
//   caller does not ask for per-sample triangle ids: a retiring lane re-reads its hit triangle's record -- a line its own traversal
//   fetched a few iterations ago, where the shading pass a millisecond later would find it evicted -- and parks (hit position, emitter
//   ordinal | -2 for a miss); otherwise (u, v, leaf slot).  Same hit_position(), same bits.  Round 5: +1.3 % with the shading pass taking
//   two rounds of a pixel side by side (EXPERIMENTS.md).
//   The slots are only ever exchanged between waves of THIS workgroup, so __syncthreads() orders them (the waves of a workgroup
//   share their CU's write-through L1; an agent-scope __threadfence() here flushes that L1 -- including the hot upper BVH levels --
//   once per tile and was measured 9 % slower per fence pair).
template <bool SPEC, bool COUNT, int LAYOUT, int TILE_STACK>
__device__ __forceinline__ void tile_body(const BakeArgs& a, int64_t p0, int np, float4* res, uint16_t* s_sorted, uint32_t* s_stack, int* s_chunk,
                                          uint32_t* ovf, TraceStats& ts, uint32_t& n_rays) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int spp = a.spp;
    float2* res_g = reinterpret_cast<float2*>(res + kTileRays);   // GGX weights (g1, g0): second array of the slab
    int lpp, ppw, rounds;
    reduce_geometry(spp, lpp, ppw, rounds);
    const int sub = lane / lpp, sl = lane - sub * lpp;
    const float inv_spp = 1.0f / (float)spp;
    const int nr = np * spp;
#if IRIS_RESOLVE_AT_RETIRE
    const bool resolve = !a.em.emit_ord && !a.tri_next;      // fused records, and nobody asks for the per-sample triangle ids
#endif

    // r / spp for a tile-local ray index without the 25-instruction integer division: exact for r * spp < 2^32 (r < kTileRays, spp <= kTileRays)
    const uint32_t spp_m = spp > 1 ? (uint32_t)(0x100000000ull / (uint64_t)(uint32_t)spp) + 1u : 0u;
    auto div_spp = [&](int r) -> int { return spp > 1 ? (int)__umulhi((uint32_t)r, spp_m) : r; };

    // The tangent frame of a pixel (utils/ops.py:12-30: a cross product, a normalisation with its square root and three divisions) is the same for
    // all of its spp samples: computed once per pixel into the part of the stack region the sort does not use (room for kFramePx pixels, i.e. every
    // tile of spp >= 40), instead of once per ray in phase A.  Same function, same bits.
    constexpr int kFrameOff = (kTileRays + 2 * 256 * 4) / 4, kFramePx = (TILE_STACK * kBlock - kFrameOff) / 6;
    float* s_frames = reinterpret_cast<float*>(s_stack) + kFrameOff;
    const bool frames = np <= kFramePx;
    if (frames) {
        for (int pl = tid; pl < np; pl += kBlock) {
            f3 t, b;
            normal_space(ld3(a.nrm + (p0 + pl) * 3), t, b);
            float* f = s_frames + pl * 6;
            f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = b.x; f[4] = b.y; f[5] = b.z;
        }
        __syncthreads();
    }

    // phases A-C (iris_tile.h): sample every ray (uniforms -> direction + GGX weights) and park it; sort by direction; trace
    tile_sort_trace<LAYOUT, COUNT, kTileRays, TILE_STACK, true, (IRIS_PARK != 0)>(
        a.sc, nr, s_sorted, s_stack, s_chunk, ovf, ts,
        [&](int r) -> uint32_t {
            const int pl = div_spp(r), s = r - pl * spp;
            const int64_t p = p0 + pl;
            const f3 n = ld3(a.nrm + p * 3), w = SPEC ? ld3(a.wo + p * 3) : mk3(0.f, 0.f, 1.f);
            const uint64_t base = (uint64_t)(a.pix_id ? (int64_t)a.pix_id[p] : p) * (uint64_t)spp;
            f3 t, b;
            if (frames) { const float* f = s_frames + pl * 6; t = mk3(f[0], f[1], f[2]); b = mk3(f[3], f[4], f[5]); }
            else normal_space(n, t, b);
            f3 wi; float g0, g1;
#if IRIS_EXP_NOSAMPLE
            // (UPPER-BOUND EXPERIMENT, wrong results: what does the sampling arithmetic cost?  A direction from three multiplies instead of Philox + the samplers; never shipped)
            { const float fs = (float)(s & 15) * 0.0625f - 0.47f, ft = (float)(s >> 4) * 0.125f - 0.45f;
              wi = t_normalize(mk3(n.x + fs * t.x + ft * b.x, n.y + fs * t.y + ft * b.y, n.z + fs * t.z + ft * b.z)); g0 = 1.f; g1 = 0.5f; (void)base; (void)w; }
#else
            sample_lobe<SPEC>(a, p, s, n, w, t, b, base, wi, g0, g1);
#endif
            res[r] = make_float4(wi.x, wi.y, wi.z, 0.f);
            if (SPEC) res_g[r] = make_float2(g1, g0);
            return dir_bin(wi);
        },
        [&](int r, f3& o, f3& d) {
            const int64_t p = p0 + div_spp(r);
            o = ld3(a.pos + p * 3);                       // raw: the pixel's position; prepare() offsets it
            const float4 qa = res[r];
            d = mk3(qa.x, qa.y, qa.z);
        },
        // position + RayEpsilon*wi (bake_shading.py:117, :180)
        [&](f3& o, f3& d) { o = mk3(o.x + kRayEps * d.x, o.y + kRayEps * d.y, o.z + kRayEps * d.z); },
        [&](int r, const Hit& h) {
#if IRIS_RESOLVE_AT_RETIRE
            if (resolve) {
                // the hit is resolved while its record is still near: position + emitter ordinal instead of (u, v, leaf slot); w = -2: a miss
                float4 q = make_float4(0.f, 0.f, 0.f, __int_as_float(-2));
                if (h.slot >= 0) {
                    const float4* tr = a.sc.tris + (int64_t)h.slot * 4;
                    const float4 tx = tr[0], ty = tr[1], tz = tr[2];
                    const float ow = tr[3].x;
                    const f3 pn = hit_position(h, mk3(tx.x, ty.x, tz.x), mk3(tx.y, ty.y, tz.y), mk3(tx.z, ty.z, tz.z));
                    q = make_float4(pn.x, pn.y, pn.z, ow);
                }
                res[r] = q;
            } else
#endif
            res[r] = make_float4(h.u, h.v, __int_as_float(h.slot), 0.f);
            if (COUNT) n_rays++; }   // every ray retires once
#if IRIS_PARK
        , a.park + (size_t)blockIdx.x * (kBlock / 64) * kParkCap * kParkWords4
#endif
        );

    // ---- phase D: shade every sample (hit -> p_next -> eval_emitter(p_next, wi, tri_next, ones, trace_roughness=0.0),
    // bake_shading.py:121-122, :184-185 -> Le * g) and take the per-pixel mean in the fixed order of the pixel-per-wave kernel
    // (lane-strided partial sums, xor butterfly)
    const int n_groups = (np + ppw - 1) / ppw;
    IRIS_PHASE_BEGIN();
    for (int g = wave; g < n_groups; g += kBlock / 64) {
        const int pl = g * ppw + sub;
        const bool pvalid = pl < np;
        float a0x = 0.f, a0y = 0.f, a0z = 0.f, a1x = 0.f, a1y = 0.f, a1z = 0.f;
#if IRIS_RESOLVE_AT_RETIRE
        // Resolved slots (position + emitter ordinal): what is left of eval_emitter is a chain of three dependent loads per sample -- slot -> voxel index -> radiance row --
        // with four live registers per sample in flight, so TWO rounds of a pixel run side by side, stage by stage (the sums take them in the order of the plain loop).
        if (resolve) {
            for (int rr = 0; rr < rounds; rr += 2) {
                const int s0 = rr * 64 + sl, s1 = s0 + 64;
                const bool ok0 = pvalid && s0 < spp, ok1 = pvalid && rr + 1 < rounds && s1 < spp;
                float4 q0 = make_float4(0.f, 0.f, 0.f, __int_as_float(-2)), q1 = q0;
                float2 w0 = make_float2(0.f, 0.f), w1 = w0;
                if (ok0) { q0 = res[pl * spp + s0]; if (SPEC) w0 = res_g[pl * spp + s0]; }
                if (ok1) { q1 = res[pl * spp + s1]; if (SPEC) w1 = res_g[pl * spp + s1]; }
                const int o0 = __float_as_int(q0.w), o1 = __float_as_int(q1.w);          // -2 miss, -1 a surface, >= 0 an emitter
                int j0 = -1, j1 = -1;
                if (o0 == -1) j0 = slf_index(a.slf, mk3(q0.x, q0.y, q0.z));
                if (o1 == -1) j1 = slf_index(a.slf, mk3(q1.x, q1.y, q1.z));
                float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
                if (o0 >= 0 || j0 >= 0) r0 = *(o0 >= 0 ? a.em.radiance + o0 : a.slf.radiance + j0);
                if (o1 >= 0 || j1 >= 0) r1 = *(o1 >= 0 ? a.em.radiance + o1 : a.slf.radiance + j1);
                if (ok0) {
                    const float z = o0 >= 0 ? -0.f : 0.f;                                  // (eval_emitter1: the cache's radiance is ADDED to a zero, the emitter's is taken)
                    const f3 Le = mk3(z + r0.x, z + r0.y, z + r0.z);
                    if (a.src_next) a.src_next[(p0 + pl) * spp + s0] = o0 >= 0 ? -2 - o0 : j0;
                    if (SPEC) { a0x += Le.x * w0.y; a0y += Le.y * w0.y; a0z += Le.z * w0.y; a1x += Le.x * w0.x; a1y += Le.y * w0.x; a1z += Le.z * w0.x; }
                    else { a0x += Le.x; a0y += Le.y; a0z += Le.z; }
                }
                if (ok1) {
                    const float z = o1 >= 0 ? -0.f : 0.f;
                    const f3 Le = mk3(z + r1.x, z + r1.y, z + r1.z);
                    if (a.src_next) a.src_next[(p0 + pl) * spp + s1] = o1 >= 0 ? -2 - o1 : j1;
                    if (SPEC) { a0x += Le.x * w1.y; a0y += Le.y * w1.y; a0z += Le.z * w1.y; a1x += Le.x * w1.x; a1y += Le.y * w1.x; a1z += Le.z * w1.x; }
                    else { a0x += Le.x; a0y += Le.y; a0z += Le.z; }
                }
            }
        } else
#endif
        for (int rr = 0; rr < rounds; ++rr) {
            const int s = rr * 64 + sl;
            if (pvalid && s < spp) {
                const float4 qa = res[pl * spp + s];
#if IRIS_EXP_NOSHADE
                // (UPPER-BOUND EXPERIMENT, wrong results: what does the shading pass cost?  The hit record is summed as it is; never shipped)
                a0x += qa.x; a0y += qa.y; a0z += qa.z; if (SPEC) { a1x += qa.x; a1y += qa.y; a1z += qa.z; }
                continue;
#endif
                Hit h; h.u = qa.x; h.v = qa.y; h.slot = __float_as_int(qa.z); h.t = 0.f; h.id = 0;
                f3 pn = mk3(0.f, 0.f, 0.f);
                int64_t tri = -1;
                int ord_rec = -1;
                if (h.slot >= 0) {
                    const float4* tr = a.sc.tris + (int64_t)h.slot * 4;
                    const float4 tx = tr[0], ty = tr[1], tz = tr[2];      // component-major record (iris_trace.h)
                    // (fused records, a.em.emit_ord == NULL, wave-uniform: the record's fourth plane carries the triangle's emitter ordinal -- a fourth load from the line the
                    //  other three come from instead of a gather into the 4 MB ordinal table: one scattered L2 request per sample less)
                    if (!a.em.emit_ord) ord_rec = __float_as_int(tr[3].x);
                    pn = hit_position(h, mk3(tx.x, ty.x, tz.x), mk3(tx.y, ty.y, tz.y), mk3(tx.z, ty.z, tz.z));
                    tri = __float_as_int(tx.w);
                }
                if (a.tri_next) a.tri_next[(p0 + pl) * spp + s] = tri;
                float epdf; bool vn; int src;
                const f3 Le = eval_emitter1(a.em, a.slf, pn, tri, true, 1.0f, 0.0f, epdf, vn, src, ord_rec);
                if (a.src_next) a.src_next[(p0 + pl) * spp + s] = src;
                if (SPEC) {
                    const float2 qb = res_g[pl * spp + s];   // (g1, g0)
                    a0x += Le.x * qb.y; a0y += Le.y * qb.y; a0z += Le.z * qb.y;
                    a1x += Le.x * qb.x; a1y += Le.y * qb.x; a1z += Le.z * qb.x;
                } else { a0x += Le.x; a0y += Le.y; a0z += Le.z; }
            }
        }
        for (int m = 1; m < lpp; m <<= 1) {
            a0x += __shfl_xor(a0x, m); a0y += __shfl_xor(a0y, m); a0z += __shfl_xor(a0z, m);
            if (SPEC) { a1x += __shfl_xor(a1x, m); a1y += __shfl_xor(a1y, m); a1z += __shfl_xor(a1z, m); }
        }
        if (pvalid && sl == 0) {
            const int64_t p = p0 + pl;
            st3(a.out0 + p * 3, mk3(a0x * inv_spp, a0y * inv_spp, a0z * inv_spp));
            if (SPEC) st3(a.out1 + p * 3, mk3(a1x * inv_spp, a1y * inv_spp, a1z * inv_spp));
        }
    }
    IRIS_PHASE_MARK(3);
}

// XCD-aware tile queue.  Workgroups are dealt round-robin to the 8 XCDs (blockIdx & 7), each with its own L2: every XCD draws from its
// own interleaved set of 64-tile chunks (neighbouring tiles = neighbouring pixels = the same BVH neighbourhood in that XCD's L2) and
// steals from the other XCDs' sets when its own is dry.  counters: 8 zeroed uints.  Returns n_tiles when nothing is left.
// Measured +1.2 % against a single global counter; results do not depend on which workgroup takes which tile.
constexpr int kTileChunk = 64;   // tiles per chunk of the queue
__device__ __forceinline__ long long claim_tile(unsigned int* counters, long long n_tiles) {
    constexpr int kChunk = kTileChunk;
    const long long n_chunks = (n_tiles + kChunk - 1) / kChunk;
    for (int k = 0; k < 8; ++k) {
        const int x = ((int)blockIdx.x + k) & 7;
        const long long per = (n_chunks + 7 - x) / 8;                 // chunks owned by XCD x: x, x + 8, ...
        for (;;) {
            const unsigned int c = atomicAdd(counters + x, 1u);
            const long long ch = (long long)(c / kChunk);
            if (ch >= per) break;                                     // this set is exhausted: steal from the next one
            const long long t = (ch * 8 + x) * kChunk + (c % kChunk);
            if (t < n_tiles) return t;                                // (the last chunk may be partial)
        }
    }
    return n_tiles;
}

// The instrumented (COUNT) instantiations are compiled for 4 waves/SIMD instead of IRIS_TILE_WAVES: the counters need ~25 more live
// registers, and at 72 VGPRs hipcc 7.2 spilled 73 of them and (round 1, commit 6b44483) produced stack-depth counters that differed
// from the pixel-per-wave kernel's -- per-ray quantities that cannot depend on the schedule (tests/test_stats.py asserts their equality).
// Counts do not depend on occupancy; the grid and the workspace stay those of the production kernel.
template <bool SPEC, bool COUNT, int LAYOUT>
__global__ __launch_bounds__(kBlock, COUNT ? 4 : IRIS_TILE_WAVES) void bake_tile_kernel(BakeArgs a) {
    constexpr int kTileStack = IRIS_TILE_STACK;  // 10240 B ray list + kTileStack KiB stacks + 12 B must fit 160 KiB / IRIS_TILE_WAVES
    __shared__ uint16_t s_sorted[kTileRays];
    __shared__ uint32_t s_stack[kTileStack * kBlock];
    __shared__ int s_tile, s_chunk;
    static_assert(kTileStack * kBlock * 4 >= kTileRays + 2 * 256 * 4, "stack region too small to alias the sort keys");
    const int tid = threadIdx.x;
    constexpr int NC = SPEC ? 2 : 1;
    float4* res = a.scratch + (size_t)blockIdx.x * kTileRays * NC;
    uint32_t* ovf = a.stack_ovf + (size_t)blockIdx.x * (kStackCapacity - kTileStack) * kBlock;   // wave-uniform (Stack adds the lane)
    const int64_t n_tiles = (a.P + a.tile_px - 1) / a.tile_px;
    TraceStats ts;
    uint32_t n_rays = 0;
    for (;;) {
        __syncthreads();  // previous tile fully done with LDS
        if (tid == 0) { s_tile = (int)claim_tile(a.tile_counter, n_tiles); s_chunk = 0; }
        (s_stack + kTileRays / 4)[tid] = 0;  // histogram: kBlock == 256 bins
        __syncthreads();
        const int64_t tile = s_tile;
        if (tile >= n_tiles) break;
        const int64_t p0 = (int64_t)tile * a.tile_px;
        tile_body<SPEC, COUNT, LAYOUT, kTileStack>(a, p0, (int)min((int64_t)a.tile_px, a.P - p0), res, s_sorted, s_stack, &s_chunk, ovf, ts, n_rays);
    }
    flush_stats<COUNT>(a, ts, n_rays);
}

// ------------------------------------------------------------------------------------------------------- view kernel
// bake_view_kernel: ALL lobes of a view (diffuse + the specular roughness levels, bake_shading.py:93-204) behind ONE persistent
// launch and ONE tile queue.  Per tile it is exactly bake_tile_kernel (tile_body, same bits); what it removes is the idle
// time at the end of every per-lobe launch, when the last ~3 ms tiles run on a few CUs -- which matters once a view is sharded
// over 8 GPUs and a rank has only ~3 tiles per resident workgroup per lobe.
constexpr int kMaxLobes = 8;
// The queue of a view runs over VIRTUAL tiles: the valid pixels are cut into spans of span_px pixels (= kTileChunk tiles of the lobe with the smallest
// tiles); chunk g of the queue (kTileChunk virtual tiles) is lobe (g / 8) % n_lobes of span ((g / 8) / n_lobes) * 8 + g % 8.  claim_tile() deals chunk g
// to XCD g % 8, so an XCD works through ALL lobes of one span -- ~2000 neighbouring pixels, whose rays start in the same corner of the BVH -- before it
// moves to its next span, and its L2 keeps that neighbourhood for seven lobes instead of one (measured +0.8 % against lobe-major order, in which an
// XCD's consecutive chunks are spans 8 apart of the same lobe; chunks of 16 / 32 / 128 tiles: +0.3 / +0.8 / +0.6 %).  A lobe with larger tiles
// fills only the first tiles_per_span slots of its chunks; the empty slots (and those behind the last pixel) are claimed and skipped.
struct ViewLobe { float rough; int spp; uint32_t stream_id; int spec; int tile_px; int tiles_per_span; float* out0; float* out1; };
struct ViewArgs {
    BakeArgs base;          // scene / tables / pixel tensors / seed / scratch / tile_counter (per-lobe fields unused)
    int n_lobes;
    int span_px;
    long long n_tiles;      // virtual tiles: spans (rounded up to 8) x lobes x kTileChunk
    ViewLobe lobe[kMaxLobes];
};

// Same occupancy as the tile kernel: 7 waves per SIMD with 10-entry LDS stacks (10240 + 10240 + 8 B of LDS per workgroup).
template <int LAYOUT>
__global__ __launch_bounds__(kBlock, IRIS_TILE_WAVES) void bake_view_kernel(ViewArgs v) {
    constexpr int kTileStack = IRIS_TILE_STACK;
    __shared__ uint16_t s_sorted[kTileRays];
    __shared__ uint32_t s_stack[kTileStack * kBlock];
    __shared__ int s_tile, s_chunk;
    static_assert(kTileStack * kBlock * 4 >= kTileRays + 2 * 256 * 4, "stack region too small to alias the sort keys");
    const int tid = threadIdx.x;
    float4* res = v.base.scratch + (size_t)blockIdx.x * kTileRays * 2;
    uint32_t* ovf = v.base.stack_ovf + (size_t)blockIdx.x * (kStackCapacity - kTileStack) * kBlock;   // wave-uniform: the traversal stacks never go to private scratch (register spills outside the node / leaf loops do: make resource-usage)
    for (;;) {
        __syncthreads();
        if (tid == 0) { s_tile = (int)claim_tile(v.base.tile_counter, v.n_tiles); s_chunk = 0; }
        (s_stack + kTileRays / 4)[tid] = 0;   // histogram
        __syncthreads();
        const long long vt = s_tile;
        if (vt >= v.n_tiles) break;
        const long long g = vt / kTileChunk, q = g >> 3;
        const int j = (int)(vt % kTileChunk), l = (int)(q % v.n_lobes);                 // (all wave-uniform)
        const int64_t span0 = ((q / v.n_lobes) * 8 + (g & 7)) * (int64_t)v.span_px;
        const int64_t p0 = span0 + (int64_t)j * v.lobe[l].tile_px;
        if (j >= v.lobe[l].tiles_per_span || p0 >= v.base.P) continue;                  // an empty slot of the virtual numbering
        const int np = (int)min((int64_t)v.lobe[l].tile_px, min(v.base.P, span0 + v.span_px) - p0);
        BakeArgs a = v.base;
        a.spp = v.lobe[l].spp; a.rough = v.lobe[l].rough; a.stream_id = v.lobe[l].stream_id; a.tile_px = v.lobe[l].tile_px;
        a.out0 = v.lobe[l].out0; a.out1 = v.lobe[l].out1; a.u2 = nullptr; a.tri_next = nullptr; a.src_next = nullptr;
        TraceStats ts; uint32_t n_rays = 0;   // unused (COUNT = false)
        if (v.lobe[l].spec) tile_body<true, false, LAYOUT, kTileStack>(a, p0, np, res, s_sorted, s_stack, &s_chunk, ovf, ts, n_rays);
        else tile_body<false, false, LAYOUT, kTileStack>(a, p0, np, res, s_sorted, s_stack, &s_chunk, ovf, ts, n_rays);
    }
}

}  // namespace iris
