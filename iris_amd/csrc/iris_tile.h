// Tile machinery shared by the bake kernels (iris_bake.h) and the large-batch path-tracing stages (iris_pt.h):
// a 256-thread workgroup takes a tile of <= CAP rays, bins them by direction with an LDS counting sort and traces them in sorted
// order with persistent lanes (trace_stream).  What a "ray" is -- how it is sampled, where its direction is parked, what happens
// with the hit -- is supplied by the caller as functors.
#pragma once
#include "iris_trace.h"

namespace iris {

// Direction bin: octahedral map of the unit vector to [0,1)^2, 16x16 cells, Morton-interleaved (adjacent codes = adjacent cones)
__device__ __forceinline__ uint32_t dir_bin(f3 d) {
    float inv = 1.0f / (fabsf(d.x) + fabsf(d.y) + fabsf(d.z) + 1e-30f);
    float px = d.x * inv, py = d.y * inv;
    if (d.z < 0.f) {
        float qx = (1.f - fabsf(py)) * (px >= 0.f ? 1.f : -1.f);
        float qy = (1.f - fabsf(px)) * (py >= 0.f ? 1.f : -1.f);
        px = qx; py = qy;
    }
    int ix = min(15, max(0, (int)((px * 0.5f + 0.5f) * 16.f)));
    int iy = min(15, max(0, (int)((py * 0.5f + 0.5f) * 16.f)));
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) m |= (((uint32_t)ix >> k) & 1u) << (2 * k) | (((uint32_t)iy >> k) & 1u) << (2 * k + 1);
    return m;
}

// Copies the first n nodes (64-B records, breadth-first order: the top levels of the tree) into LDS at an 80-B stride.  Once per
// persistent workgroup; the caller's first barrier publishes them.
__device__ __forceinline__ void stage_top_nodes(const SceneDev& sc, uint4* s_top, int n) {
    const int have = min(n, sc.n_nodes);
    for (int i = threadIdx.x; i < n * 4; i += kBlock) {
        const int node = i >> 2, q = i & 3;
        // nodes the tree does not have are never referenced; fill them with an all-miss node anyway
        s_top[node * kLdsNodeQuads + q] = node < have ? reinterpret_cast<const uint4*>(sc.nodes)[i] : make_uint4(0u, 0u, 0xffffffffu, q == 3 ? 0xffffffffu : 0u);
    }
}

// LDS contract: s_sorted[CAP] (uint16 ray list), s_stack[TILE_STACK * 256] (traversal stacks; doubles as the sort's key / histogram /
// cursor storage: CAP bytes of keys, then 256 + 256 words -- the uses are separated by workgroup barriers), *s_chunk (cursor).
// Before the call the caller has zeroed the histogram (s_stack + CAP/4, 256 words) and *s_chunk and passed a barrier.
//   phase_a(r) -> direction bin : sample ray r of the tile and park whatever phase C / the caller's epilogue need
//   fetch_ray(r, o, d)          : ISSUE the loads of ray r's raw origin / direction (no dependent arithmetic)
//   prepare(o, d)               : raw -> actual origin / direction (first use of the loaded values)
//   retire(r, h)                : store the hit of ray r
// Everything exchanged through global memory here stays inside ONE workgroup, so __syncthreads() orders it (the waves of a
// workgroup share their CU's write-through L1; an agent-scope __threadfence() would flush that L1 -- including the hot upper BVH
// levels -- once per tile and was measured 9 % slower per fence pair).  Ends with a barrier: hits are visible to the caller.
//   s_top (LDS_NODES > 0)       : the first LDS_NODES nodes staged in LDS by stage_top_nodes()
//   tail (MERGE_TAIL)           : LDS for the merged tail: when the list is exhausted a wave that is down to <= kTailMax unfinished rays
//                                 parks their traversal state here (best hit, current reference, stack depth; the stack contents stay
//                                 in the parking lane's column) and leaves; after a barrier wave 0 adopts all parked rays of the tile
//                                 (<= 4 x kTailMax = 64): each adopting lane copies the parked lane's stack column into its own and
//                                 continues the traversal where it stopped -- one wave's tail instead of four, nothing is re-traversed.
struct TileTail {
    float4 hit[4 * kTailMax];      // (t, u, v, leaf slot)
    uint32_t cur[4 * kTailMax];    // node / leaf reference the ray was at
    int32_t id[4 * kTailMax];      // triangle id of the best hit (tie-break)
    uint32_t meta[4 * kTailMax];   // ray id | stack depth << 16 | parking thread << 24
    int n;
};
template <int LAYOUT, bool COUNT, int CAP, int TILE_STACK, bool GLOBAL_OVF, int LDS_NODES, bool MERGE_TAIL, class PhaseA, class FetchRay, class Prepare, class Retire>
__device__ __forceinline__ void tile_sort_trace(const SceneDev& sc, int nr, uint16_t* s_sorted, uint32_t* s_stack, int* s_chunk, uint32_t* ovf,
                                                const uint4* s_top, TileTail* tail, TraceStats& ts, PhaseA phase_a, FetchRay fetch_ray, Prepare prepare,
                                                Retire retire) {
    static_assert(TILE_STACK * kBlock * 4 >= CAP + 2 * 256 * 4, "stack region too small to alias the sort keys");
    uint8_t* s_keys = reinterpret_cast<uint8_t*>(s_stack);
    uint32_t* s_hist = s_stack + CAP / 4;
    uint32_t* s_cur = s_hist + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (MERGE_TAIL && tid == 0) tail->n = 0;     // (published by the barriers below, long before the first wave can park)
    // ---- phase A: sample, park, histogram of the direction bins
    for (int r = tid; r < nr; r += kBlock) {
        const uint32_t key = phase_a(r);
        s_keys[r] = (uint8_t)key;
        atomicAdd(&s_hist[key], 1u);
    }
    __syncthreads();
    // ---- exclusive prefix over the 256 bins (wave 0: 4 bins per lane)
    if (wave == 0) {
        uint32_t c0 = s_hist[lane * 4], c1 = s_hist[lane * 4 + 1], c2 = s_hist[lane * 4 + 2], c3 = s_hist[lane * 4 + 3];
        uint32_t tot = c0 + c1 + c2 + c3, inc = tot;
        for (int m = 1; m < 64; m <<= 1) { uint32_t v = __shfl_up(inc, m); if (lane >= m) inc += v; }
        uint32_t ex = inc - tot;
        s_cur[lane * 4] = ex; s_cur[lane * 4 + 1] = ex + c0; s_cur[lane * 4 + 2] = ex + c0 + c1; s_cur[lane * 4 + 3] = ex + c0 + c1 + c2;
    }
    __syncthreads();
    // ---- phase B: scatter ray ids into bin order (order inside a bin is irrelevant: hits go to per-ray slots)
    for (int r = tid; r < nr; r += kBlock) {
        const uint32_t pos = atomicAdd(&s_cur[s_keys[r]], 1u);
        s_sorted[pos] = (uint16_t)r;
    }
    __syncthreads();  // keys / histogram dead from here on: the region becomes the traversal stacks
    // ---- phase C: persistent-lane traversal of the sorted list: idle lanes claim the next rays together
    {
        int my_r = 0;
        auto fetch = [&](f3& o, f3& d) -> bool {
            const unsigned long long m = __ballot(1);
            int base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(s_chunk, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            const int i = base + __popcll(m & ((1ull << lane) - 1ull));
            if (i >= nr) return false;
            my_r = s_sorted[i];
            fetch_ray(my_r, o, d);
            return true;
        };
        auto ret = [&](const Hit& h) { retire(my_r, h); };
        auto park = [&](const RayState& r, const Stack<TILE_STACK, GLOBAL_OVF, LDS_NODES>& st) {
            const unsigned long long m = __ballot(1);
            int base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&tail->n, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            const int i = base + __popcll(m & ((1ull << lane) - 1ull));
            tail->hit[i] = make_float4(r.h.t, r.h.u, r.h.v, __int_as_float(r.h.slot));
            tail->cur[i] = r.cur;
            tail->id[i] = r.h.id;
            tail->meta[i] = (uint32_t)my_r | (uint32_t)st.sp << 16 | (uint32_t)tid << 24;
        };
        trace_stream<LAYOUT, COUNT, TILE_STACK, GLOBAL_OVF, LDS_NODES, MERGE_TAIL ? kTailMax : 0>(
            sc, s_stack + tid, ovf, s_top, &ts, fetch, prepare, ret, [](RayState&, Stack<TILE_STACK, GLOBAL_OVF, LDS_NODES>&) {}, park);
    }
    __syncthreads();
    if (MERGE_TAIL) {
        // ---- merged tail: the <= 64 parked rays of the tile, adopted by wave 0
        const int n_tail = tail->n;
        if (n_tail > 0 && wave == 0) {
            int my_r = 0;
            bool first = true;
            uint32_t meta = 0;
            auto fetch = [&](f3& o, f3& d) -> bool {
                if (!first || lane >= n_tail) { first = false; return false; }
                first = false;
                meta = tail->meta[lane];
                my_r = (int)(meta & 0xffffu);
                fetch_ray(my_r, o, d);
                return true;
            };
            auto resume = [&](RayState& r, Stack<TILE_STACK, GLOBAL_OVF, LDS_NODES>& st) {
                const float4 ph = tail->hit[lane];
                r.h.t = ph.x; r.h.u = ph.y; r.h.v = ph.z; r.h.slot = __float_as_int(ph.w); r.h.id = tail->id[lane];
                r.cur = tail->cur[lane];
                const int sp = (int)((meta >> 16) & 0xffu), src = (int)(meta >> 24);
                // copy the parking lane's stack column (level by level: every lane reads before any lane writes, so columns of wave 0
                // that are both source and destination are safe)
                for (int k = 0; k < sp; ++k) {
                    uint32_t v;
                    if (k < TILE_STACK) v = s_stack[k * kBlock + src];
                    else v = ovf[(uint32_t)min(k - TILE_STACK, kStackCapacity - TILE_STACK - 1) * kBlock + src];
                    st.push(v);
                }
            };
            auto ret = [&](const Hit& h) { retire(my_r, h); };
            trace_stream<LAYOUT, COUNT, TILE_STACK, GLOBAL_OVF, LDS_NODES, 0>(sc, s_stack + tid, ovf, s_top, &ts, fetch, prepare, ret, resume,
                                                                              [](const RayState&, const Stack<TILE_STACK, GLOBAL_OVF, LDS_NODES>&) {});
        }
        __syncthreads();
    }
}

}  // namespace iris
