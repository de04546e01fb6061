// Streamed bake of a whole view (iris_bake_view): the same per-ray arithmetic and the same per-pixel reduction order as the tile kernels
// (iris_bake.h: identical bits), but the three phases of a tile are three kernels over coalesced ray / hit buffers in HBM, so that the
// traversal is no longer tied to tiles:
//
//   stream_sample_kernel   one workgroup per tile (<= kTileRays rays of one lobe): uniforms -> BRDF sample (+ GGX weights) -> direction bin ->
//                          LDS counting sort -> the tile's rays as 16-B RECORDS {wi, ray-in-tile | pixel-in-tile << 16} in direction order
//                          (rec[tile * kTileRays + rank]); weights to gw[ray]; first pixel of the tile to tile_p0[tile]
//   stream_trace_kernel    persistent waves, no workgroup barriers: every wave draws SEGMENTS (kSegRays consecutive records = a contiguous
//                          direction range of one tile) from an XCD-aware queue and refills its idle lanes from the segment -- and, when the
//                          segment is exhausted, straight from the next one: there is no per-tile drain (9 % of the node iterations of the
//                          tile kernel ran at 4 % lane utilisation); a finished ray is shaded at once (emitter / SLF lookup) and its
//                          radiance goes to hit[ray]
//   stream_shade_kernel    per pixel: Le * g, fixed-order sum (reduce_geometry) -> output maps (a pure stream over the radiance / weight slots)
//
// The sampling and shading passes are HBM-bound (40 B per ray of buffers + the shading gathers), the traversal is VALU-bound:
// the host runs chunk k's traversal beside chunk k+1's sampling and chunk k-1's shading (two streams, two buffer sets; the traversal is
// launched one workgroup per CU short of what it is compiled for, which leaves the other passes a wave slot and registers on every SIMD).
// Unused slots of a partial tile hold a null ray (pixel -1: origin 1e30, misses the root in one node step).
#pragma once
#include "iris_bake.h"

namespace iris {

constexpr int kSegRays = 1280;                       // records per traversal segment (a quarter of a full tile's sorted list)
constexpr long long kStreamChunkRays = 1ll << 26;    // ray slots per chunk and buffer set (2.7 GB); a 1080p x SPP 128 x 7-lobe view is 28 chunks

#ifndef IRIS_STREAM_WAVES      // waves per SIMD the streamed traversal is compiled for: 64 VGPRs (2 spills), LDS only the 10 KiB of stacks
#define IRIS_STREAM_WAVES 8
#endif
#ifndef IRIS_STREAM_BLOCKS     // traversal workgroups launched per CU: one less, so that a sampling / shading workgroup fits beside them
#define IRIS_STREAM_BLOCKS 7
#endif

struct StreamArgs {
    ViewArgs v;               // lobes, pixel tensors, tables (base.scratch / tile_counter unused here)
    long long tile0, n_tiles; // this chunk: tiles [tile0, tile0 + n_tiles) of the view's tile sequence
    float4* rec;              // n_tiles * kTileRays      (wi.xyz, ray-in-tile | pixel-in-tile << 16), direction-sorted per tile
    float4* hit;              // n_tiles * kTileRays      Le of the ray's hit (eval_emitter)
    float2* gw;               // n_tiles * kTileRays      (g1, g0)
    int32_t* tile_p0;         // n_tiles                  first pixel of the tile (-1: unused)
    unsigned int* seg_counter;// 8 zeroed counters (claim_tile)
    uint32_t* stack_ovf;
};

__device__ __forceinline__ int lobe_of_tile(const ViewArgs& v, long long gt) {
    int l = 0;
    for (int k = 1; k < v.n_lobes; ++k) if (gt >= v.lobe[k].tile_begin) l = k;
    return l;
}

// ------------------------------------------------------------------------------------------------------- sample + sort
template <bool SPEC>
__device__ __forceinline__ void stream_sample_tile(const BakeArgs& a, long long tile, float4* rec, float2* gw, int32_t* tile_p0, float* s_wi, uint8_t* s_keys,
                                                   uint32_t* s_hist, uint32_t* s_cur) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int spp = a.spp;
    const int64_t p0 = tile * a.tile_px;
    const int np = (int)min((int64_t)a.tile_px, a.P - p0);
    const int nr = np * spp;
    if (tid == 0) *tile_p0 = (int32_t)p0;
    // phase A: sample every ray, keep wi in LDS (3 planes of kTileRays floats) and (g1, g0) in gw, histogram of the direction bins
    for (int r = tid; r < nr; r += kBlock) {
        const int pl = r / spp, s = r - pl * spp;
        const int64_t p = p0 + pl;
        const f3 n = ld3(a.nrm + p * 3), w = SPEC ? ld3(a.wo + p * 3) : mk3(0.f, 0.f, 1.f);
        const uint64_t base = (uint64_t)(a.pix_id ? (int64_t)a.pix_id[p] : p) * (uint64_t)spp;
        f3 t, b;
        normal_space(n, t, b);
        f3 wi; float g0, g1;
        sample_lobe<SPEC>(a, p, s, n, w, t, b, base, wi, g0, g1);
        s_wi[r] = wi.x; s_wi[kTileRays + r] = wi.y; s_wi[2 * kTileRays + r] = wi.z;
        if (SPEC) gw[r] = make_float2(g1, g0);
        const uint32_t key = dir_bin(wi);
        s_keys[r] = (uint8_t)key;
        atomicAdd(&s_hist[key], 1u);
    }
    __syncthreads();
    if (wave == 0) {      // exclusive prefix over the 256 bins
        uint32_t c0 = s_hist[lane * 4], c1 = s_hist[lane * 4 + 1], c2 = s_hist[lane * 4 + 2], c3 = s_hist[lane * 4 + 3];
        uint32_t tot = c0 + c1 + c2 + c3, inc = tot;
        for (int m = 1; m < 64; m <<= 1) { uint32_t v = __shfl_up(inc, m); if (lane >= m) inc += v; }
        uint32_t ex = inc - tot;
        s_cur[lane * 4] = ex; s_cur[lane * 4 + 1] = ex + c0; s_cur[lane * 4 + 2] = ex + c0 + c1; s_cur[lane * 4 + 3] = ex + c0 + c1 + c2;
    }
    __syncthreads();
    // phase B: every ray's record at its rank (order inside a bin is irrelevant: results go to per-ray slots)
    for (int r = tid; r < nr; r += kBlock) {
        const uint32_t rank = atomicAdd(&s_cur[s_keys[r]], 1u);
        rec[rank] = make_float4(s_wi[r], s_wi[kTileRays + r], s_wi[2 * kTileRays + r], __int_as_float(r | (r / spp) << 16));
    }
    // the unused slots of a partial tile: null rays behind the real ones (pixel-in-tile 0xffff)
    for (int r = nr + tid; r < kTileRays; r += kBlock) rec[r] = make_float4(0.f, 0.f, 1.f, __int_as_float(r | 0xffff0000));
}

constexpr int kSampleLdsBytes = 3 * kTileRays * 4 + kTileRays + 2 * 256 * 4;      // wi planes + keys + histogram + cursors = 68 608 B (dynamic)
__global__ __launch_bounds__(kBlock) void stream_sample_kernel(StreamArgs A) {
    extern __shared__ float s_dyn[];
    float* s_wi = s_dyn;
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_dyn + 3 * kTileRays);
    uint32_t* s_cur = s_hist + 256;
    uint8_t* s_keys = reinterpret_cast<uint8_t*>(s_cur + 256);
    const long long j = blockIdx.x;                          // tile within the chunk
    if (j >= A.n_tiles) return;
    const long long gt = A.tile0 + j;
    const int l = lobe_of_tile(A.v, gt);
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    BakeArgs a = A.v.base;
    a.spp = A.v.lobe[l].spp; a.rough = A.v.lobe[l].rough; a.stream_id = A.v.lobe[l].stream_id; a.tile_px = A.v.lobe[l].tile_px; a.u2 = nullptr;
    float4* rec = A.rec + (size_t)j * kTileRays;
    float2* gw = A.gw + (size_t)j * kTileRays;
    if (A.v.lobe[l].spec) stream_sample_tile<true>(a, gt - A.v.lobe[l].tile_begin, rec, gw, A.tile_p0 + j, s_wi, s_keys, s_hist, s_cur);
    else stream_sample_tile<false>(a, gt - A.v.lobe[l].tile_begin, rec, gw, A.tile_p0 + j, s_wi, s_keys, s_hist, s_cur);
}

// ------------------------------------------------------------------------------------------------------- traversal
template <int LAYOUT>
__global__ __launch_bounds__(kBlock, IRIS_STREAM_WAVES) void stream_trace_kernel(StreamArgs A) {
    constexpr int kStack = IRIS_TILE_STACK;
    __shared__ uint32_t s_stack[kStack * kBlock];
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t* ovf = A.stack_ovf + (size_t)blockIdx.x * (kStackCapacity - kStack) * kBlock;
    const long long n_seg = A.n_tiles * (kTileRays / kSegRays);
    TraceStats ts;
    const BakeArgs& a = A.v.base;
    // The wave's current segment -- records [pos, end) of the chunk -- lives in LDS: fetch() runs on the idle lanes only, so wave-uniform state
    // kept in registers would go stale in the lanes that sit a refill out.  Only this wave touches its two words.
    __shared__ uint32_t s_seg_[kBlock / 64][2];
    volatile uint32_t* s_seg = &s_seg_[tid >> 6][0];     // (volatile: written by one lane, read by another lane of the same wave later)
    if (lane == 0) { s_seg[0] = 0u; s_seg[1] = 0u; }
    uint32_t my_ray = 0;
    auto fetch = [&](f3& o, f3& d) -> bool {
        const unsigned long long m = __ballot(1);
        const int rank = __popcll(m & ((1ull << lane) - 1ull)), need = __popcll(m);
        // the first requesting lane serves `need` (<= 64 < kSegRays) consecutive records from the current segment and, where that runs out,
        // from the next one it claims: two (base, count) runs
        uint32_t b0 = 0, c0 = 0, b1 = 0, c1 = 0;
        if (lane == __ffsll((long long)m) - 1) {
            uint32_t pos = s_seg[0], end = s_seg[1];
            b0 = pos; c0 = min((uint32_t)need, end - pos); pos += c0;
            if (c0 < (uint32_t)need && end != 0xffffffffu) {
                const long long sg = claim_tile(A.seg_counter, n_seg);
                if (sg < n_seg) { pos = (uint32_t)(sg * kSegRays); end = pos + kSegRays; b1 = pos; c1 = (uint32_t)need - c0; pos += c1; }
                else { pos = end = 0xffffffffu; }      // nothing left: this wave stops asking
            }
            s_seg[0] = pos; s_seg[1] = end;
        }
        b0 = __builtin_amdgcn_readfirstlane(b0); c0 = __builtin_amdgcn_readfirstlane(c0);
        b1 = __builtin_amdgcn_readfirstlane(b1); c1 = __builtin_amdgcn_readfirstlane(c1);
        uint32_t idx;
        if ((uint32_t)rank < c0) idx = b0 + (uint32_t)rank;
        else if ((uint32_t)rank < c0 + c1) idx = b1 + ((uint32_t)rank - c0);
        else return false;
        const float4 ra = A.rec[idx];
        const uint32_t tile = idx / kTileRays, code = (uint32_t)__float_as_int(ra.w), pl = code >> 16;
        d = mk3(ra.x, ra.y, ra.z);
        if (pl == 0xffffu) o = mk3(1e30f, 1e30f, 1e30f);                 // null ray
        else o = ld3(A.v.base.pos + ((int64_t)A.tile_p0[tile] + pl) * 3);   // raw: the pixel's position; prepare() offsets it
        my_ray = tile * kTileRays + (code & 0xffffu);
        return true;
    };
    trace_stream<LAYOUT, false, kStack, true>(
        a.sc, s_stack + tid, ovf, &ts, fetch,
        // position + RayEpsilon*wi (bake_shading.py:117, :180)
        [](f3& o, f3& d) { o = mk3(o.x + kRayEps * d.x, o.y + kRayEps * d.y, o.z + kRayEps * d.z); },
        // a finished ray is shaded at once, while its triangle is still in L1: hit -> p_next -> eval_emitter(p_next, wi, tri_next, ones,
        // trace_roughness=0.0) (bake_shading.py:121-122, :184-185); the radiance goes to the ray's slot, the per-pixel sums stay with the
        // reduction pass (same values, same order as the tile kernels)
        [&](const Hit& h) {
            f3 pn = mk3(0.f, 0.f, 0.f);
            int64_t tri = -1;
            if (h.slot >= 0) {
                f3 p0, p1, p2;
                hit_vertices(a.sc, h, p0, p1, p2);
                pn = hit_position(h, p0, p1, p2);
                tri = h.id;
            }
            float epdf; bool vn;
            const f3 Le = eval_emitter1(a.em, a.slf, pn, tri, true, 1.0f, 0.0f, epdf, vn);
            A.hit[my_ray] = make_float4(Le.x, Le.y, Le.z, 0.f);
        });
}

// ------------------------------------------------------------------------------------------------------- shade + reduce
template <bool SPEC>
__device__ __forceinline__ void stream_shade_tile(const BakeArgs& a, long long tile, const float4* res, const float2* res_g) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int spp = a.spp;
    int lpp, ppw, rounds;
    reduce_geometry(spp, lpp, ppw, rounds);
    const int sub = lane / lpp, sl = lane - sub * lpp;
    const float inv_spp = 1.0f / (float)spp;
    const int64_t p0 = tile * a.tile_px;
    const int np = (int)min((int64_t)a.tile_px, a.P - p0);
    // the sums of phase D of tile_body (iris_bake.h): lane-strided partial sums, xor butterfly
    const int n_groups = (np + ppw - 1) / ppw;
    for (int g = wave; g < n_groups; g += kBlock / 64) {
        const int pl = g * ppw + sub;
        const bool pvalid = pl < np;
        float a0x = 0.f, a0y = 0.f, a0z = 0.f, a1x = 0.f, a1y = 0.f, a1z = 0.f;
        for (int rr = 0; rr < rounds; ++rr) {
            const int s = rr * 64 + sl;
            if (pvalid && s < spp) {
                const float4 qa = res[pl * spp + s];
                const f3 Le = mk3(qa.x, qa.y, qa.z);
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
}

__global__ __launch_bounds__(kBlock) void stream_shade_kernel(StreamArgs A) {
    const long long j = blockIdx.x;
    if (j >= A.n_tiles) return;
    const long long gt = A.tile0 + j;
    const int l = lobe_of_tile(A.v, gt);
    BakeArgs a = A.v.base;
    a.spp = A.v.lobe[l].spp; a.rough = A.v.lobe[l].rough; a.tile_px = A.v.lobe[l].tile_px;
    a.out0 = A.v.lobe[l].out0; a.out1 = A.v.lobe[l].out1;
    const float4* res = A.hit + (size_t)j * kTileRays;
    const float2* res_g = A.gw + (size_t)j * kTileRays;
    if (A.v.lobe[l].spec) stream_shade_tile<true>(a, gt - A.v.lobe[l].tile_begin, res, res_g);
    else stream_shade_tile<false>(a, gt - A.v.lobe[l].tile_begin, res, res_g);
}

}  // namespace iris
