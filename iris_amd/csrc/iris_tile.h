// Tile machinery shared by the bake kernels (iris_bake.h) and the large-batch path-tracing stages (iris_pt.h):
// a 256-thread workgroup takes a tile of <= CAP rays, bins them by direction with an LDS counting sort and traces them in sorted
// order with persistent lanes (trace_stream).  What a "ray" is -- how it is sampled, where its direction is parked, what happens
// with the hit -- is supplied by the caller as functors.
#pragma once
#include "iris_trace.h"

namespace iris {

// Wave priorities of the phases of a tile (s_setprio, 0..3).  A CU holds 7 workgroups in different phases: the traversal waves are the ones
// waiting on memory, the sampling / shading waves are arithmetic -- letting a traversal wave issue first whenever it can gets its next loads
// out earlier, and the arithmetic of the other phases fills the gaps.  Measured (10 views): A / C / D = 0 / 0 / 0: 7.93-7.94, 0 / 3 / 0: 8.00-8.05,
// 3 / 0 / 3: 7.81; the shading phase's priority and finer levels inside the traversal (refill round, leaf loop) make no difference.
#ifndef IRIS_PRIO_A
#define IRIS_PRIO_A 0
#endif
#ifndef IRIS_PRIO_C
#define IRIS_PRIO_C 3
#endif
#ifndef IRIS_PRIO_D
#define IRIS_PRIO_D 0
#endif

// Direction bin, octant-major: key = octant (3 bits: the signs of d) | cell inside the octant (5 bits).  The BVH node table exists once per ray octant
// (iris_trace.h), so rays of one octant share node lines and visit children in the same order, and rays of different octants share nothing: a wave
// whose 64 consecutive rays of the sorted list come from one octant touches fewer lines per load.  Inside the octant the L1-normalised |d| lies in a
// triangle (a + b <= 1), mapped to the unit square by (a, b / (1 - a)) and cut into 8 x 4 cells, columns walked alternately up and down.
// Measured against the 16 x 16 Morton-ordered octahedral map of rounds 1-3 (whose cells straddle octants along the axes and diagonals): +2.0 %;
// Morton order inside the octant, 4 x 8 cells, the triangle cut directly, Gray-code order of the octants: -0.1 ... -0.8 % against this; 4 x 4 cells
// (128 bins; round 4, with the shared node visits of iris_trace.h, which like waves of one cell): -0.4 %.
// (1-ulp reciprocals are plenty for a bin: results do not depend on the binning.)
__device__ __forceinline__ uint32_t dir_bin(f3 d) {
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const float inv = __builtin_amdgcn_rcpf(ax + ay + az + 1e-30f);
    const float a = ax * inv, b = ay * inv;
    const float v = b * __builtin_amdgcn_rcpf(1.f - a + 1e-30f);
    const int iu = min(7, (int)(a * 8.f)), iv0 = min(3, (int)(v * 4.f));
    const int iv = (iu & 1) ? 3 - iv0 : iv0;
    const uint32_t oct = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
    return (oct << 5) | ((uint32_t)iu << 2) | (uint32_t)iv;
}

// Diagnostic build (-DIRIS_PHASE_TIMING, tools/diag_phases.py): shader cycles every workgroup spends in the phases of a tile, summed over
// workgroups by thread 0 (0 = A sample + bin, 1 = prefix + B scatter, 2 = C traversal, 3 = D shade + reduce).  Not compiled into the product.
#ifdef IRIS_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[8];
#define IRIS_PHASE_BEGIN() unsigned long long t_phase_ = clock64()
#define IRIS_PHASE_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_phase_cycles[k], t_ - t_phase_); t_phase_ = t_; } } while (0)
#else
#define IRIS_PHASE_BEGIN()
#define IRIS_PHASE_MARK(k)
#endif

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
template <int LAYOUT, bool COUNT, int CAP, int TILE_STACK, bool GLOBAL_OVF, bool PARK = false, class PhaseA, class FetchRay, class Prepare, class Retire>
__device__ __forceinline__ void tile_sort_trace(const SceneDev& sc, int nr, uint16_t* s_sorted, uint32_t* s_stack, int* s_chunk, uint32_t* ovf,
                                                TraceStats& ts, PhaseA phase_a, FetchRay fetch_ray, Prepare prepare, Retire retire, iris_u4v* park_pool = nullptr) {
    static_assert(TILE_STACK * kBlock * 4 >= CAP + 2 * 256 * 4, "stack region too small to alias the sort keys");
    uint8_t* s_keys = reinterpret_cast<uint8_t*>(s_stack);
    uint32_t* s_hist = s_stack + CAP / 4;
    uint32_t* s_cur = s_hist + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    IRIS_PHASE_BEGIN();
    __builtin_amdgcn_s_setprio(IRIS_PRIO_A);
    // ---- phase A: sample, park, histogram of the direction bins
    for (int r = tid; r < nr; r += kBlock) {
        const uint32_t key = phase_a(r);
        s_keys[r] = (uint8_t)key;
        atomicAdd(&s_hist[key], 1u);
    }
    __syncthreads();
    IRIS_PHASE_MARK(0);
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
    IRIS_PHASE_MARK(1);
    // ---- phase C: persistent-lane traversal of the sorted list: idle lanes claim the next rays together
    {
        int my_r = 0;
#if IRIS_PARK
        // the idle lanes of a wave claim the next rays of the sorted list together: -> the ray's id in the tile, or -1 when the list is exhausted
        auto claim = [&]() -> int {
            const unsigned long long m = __ballot(1);
            int base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(s_chunk, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            const int i = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            return i < nr ? (int)s_sorted[i] : -1;
        };
#endif
        auto fetch = [&](f3& o, f3& d) -> bool {
            const unsigned long long m = __ballot(1);
            int base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(s_chunk, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            // (v_mbcnt: the claiming lanes below this one, without the 64-bit lane mask -- a tile-long value hipcc kept in scratch and reloaded in every refill round)
            const int i = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (i >= nr) return false;
            my_r = s_sorted[i];
            fetch_ray(my_r, o, d);
            return true;
        };
        auto ret = [&](const Hit& h) { retire(my_r, h); };
        __builtin_amdgcn_s_setprio(IRIS_PRIO_C);
        // The per-lane stack addresses are derived from a FRESH copy of the thread index: as values live across the whole tile (sampling and shading
        // included) hipcc spills them when another phase needs the registers -- and then reloads them from scratch in front of every push and pop.
        uint32_t tid_c = threadIdx.x;
        asm volatile("" : "+v"(tid_c));
        // The same for the table bases (wave-uniform, scalar registers): fresh copies whose live range is the traversal -- otherwise the kernel-long values
        // are spilled to vector-register lanes and read back (v_readlane + wait states) in front of the node / triangle loads of every visit.
        SceneDev sc_c = sc;
        asm volatile("" : "+s"(sc_c.nodes), "+s"(sc_c.tris), "+s"(sc_c.oct_stride));
#if IRIS_PARK
        if constexpr (PARK && TILE_STACK >= 12) {      // (ONE instantiation of the traversal per kernel: the stage kernels of iris_pt.h do not park)
            auto get_id = [&]() -> int { return my_r; };
            auto set_id = [&](int id) { my_r = id; };
            auto refetch = [&](int id, f3& o, f3& d) { fetch_ray(id, o, d); };
            typedef __attribute__((address_space(3))) int lds_i32;        // (an LDS read: through a generic volatile pointer this is a system-coherent flat_load of a 64-bit address that gets spilled)
            auto left = [&]() -> int { return nr - __atomic_load_n((lds_i32*)s_chunk, __ATOMIC_RELAXED); };
            ParkOps<decltype(get_id), decltype(set_id), decltype(refetch), decltype(left), decltype(claim)> pk{park_pool + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kParkCap * kParkWords4, get_id, set_id, refetch, left, claim};      // (a SCALAR base: the wave index through readfirstlane)
            trace_stream<LAYOUT, COUNT, TILE_STACK, GLOBAL_OVF>(sc_c, s_stack + tid_c, tid_c, ovf, &ts, fetch, prepare, ret, pk);
        } else
#endif
            trace_stream<LAYOUT, COUNT, TILE_STACK, GLOBAL_OVF>(sc_c, s_stack + tid_c, tid_c, ovf, &ts, fetch, prepare, ret);
        __builtin_amdgcn_s_setprio(IRIS_PRIO_D);
        IRIS_PHASE_MARK(4);      // wave 0's own traversal; 2 (below) also counts its wait for the slowest wave of the tile
    }
    __syncthreads();
    IRIS_PHASE_MARK(2);
}

}  // namespace iris
