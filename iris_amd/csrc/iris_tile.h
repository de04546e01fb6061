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
//   hand (HANDOVER; GLOBAL_OVF tiles only): LDS for the hand-over of a tile's last rays between its waves (trace_stream's `tail`)
struct TileHandOver {
    float4 hit[4 * kTailMax];      // (t, u, v, leaf slot) of the best hit so far; wave w parks into entries [w * kTailMax, (w + 1) * kTailMax)
    uint32_t cur[4 * kTailMax];    // node / leaf reference the ray was at
    int32_t id[4 * kTailMax];      // triangle id of the best hit (tie-break)
    uint32_t meta[4 * kTailMax];   // ray id | stack depth << 16 | parking thread << 24
    uint32_t count[4];             // rays wave w parked (written before its bit in `state` is published)
    uint32_t taken[4];             // ... of which handed out to adopting waves
    uint32_t state;                // bits 0..7: waves of the tile still tracing; bit 8 + w: wave w has parked
};
// Protocol (all through `state`, one compare-and-swap per decision, workgroup scope): a wave parks with {running - 1, its bit set} only while
// running >= 2, so somebody is left to adopt; a wave without rays leaves with {running - 1}, but the last one (running == 1; nobody else
// can change anything any more) first checks that no parked ray is unclaimed.  The parking wave's entries, its stack columns in LDS and its
// overflow rows (global memory behind the CU's write-through L1, which the waves of a workgroup share) are written before the release.
template <int TILE_STACK, class FetchRay>
struct TileTailOps {
    static constexpr bool kEnabled = true;
    TileHandOver* t; int& my_r; FetchRay& fetch_ray; uint32_t* s_stack; uint32_t* ovf;
    bool last;                     // wave-uniform: park() was refused once -- this wave is the tile's last and stays it
    __device__ __forceinline__ uint32_t load_state() { return __hip_atomic_load(&t->state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ bool unclaimed(uint32_t s) {
        bool a = false;
#pragma unroll
        for (int w = 0; w < 4; ++w) a |= ((s >> (8 + w)) & 1u) && t->taken[w] < t->count[w];
        return a;
    }
    __device__ __forceinline__ bool available() {
        const uint32_t s = load_state();
        return (s >> 8) != 0 && unclaimed(s);
    }
    // idle lanes together: claim parked rays of ONE wave (the rest, if any, next round) and issue the loads of their raw origin / direction
    __device__ __forceinline__ bool adopt(f3& o, f3& d) {
        const int lane = threadIdx.x & 63;
        const unsigned long long m = __ballot(1);
        const int leader = __ffsll((long long)m) - 1;
        int first = -1, k = 0;
        if (lane == leader) {
            const uint32_t s = load_state();
            const int need = __popcll(m);
            for (int w = 0; w < 4 && first < 0; ++w) {
                if (!((s >> (8 + w)) & 1u)) continue;
                const uint32_t c = t->count[w];
                if (t->taken[w] >= c) continue;
                const uint32_t kk = min((uint32_t)need, c - t->taken[w]);
                const uint32_t b = atomicAdd(&t->taken[w], kk);          // another wave may be claiming too: b decides
                if (b < c) { first = w * kTailMax + (int)b; k = (int)min(kk, c - b); }
            }
        }
        first = __shfl(first, leader); k = __shfl(k, leader);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (first < 0 || rank >= k) return false;
        const int i = first + rank;
        my_r = (int)(t->meta[i] & 0xffffu) | (i + 1) << 16;               // entry kept in the ray id's upper half until resume()
        fetch_ray(my_r & 0xffff, o, d);
        return true;
    }
    template <class S> __device__ __forceinline__ void resume(RayState& r, S& st) {
        if ((my_r >> 16) == 0) return;
        const int i = (my_r >> 16) - 1;
        my_r &= 0xffff;
        const float4 ph = t->hit[i];
        r.h.t = ph.x; r.h.u = ph.y; r.h.v = ph.z; r.h.slot = __float_as_int(ph.w); r.h.id = t->id[i];
        r.cur = t->cur[i];
        const uint32_t meta = t->meta[i];
        const int sp = (int)((meta >> 16) & 0xffu), src = (int)(meta >> 24);
        for (int k = 0; k < sp; ++k) {        // the parking lane's stack column (its wave has left the traversal) into this lane's
            uint32_t v;
            if (k < TILE_STACK) v = s_stack[k * kBlock + src];
            else v = ovf[(uint32_t)min(k - TILE_STACK, kStackCapacity - TILE_STACK - 1) * kBlock + src];
            st.push(v);
        }
    }
    // whole wave: hand the unfinished rays over unless this is the last wave running
    template <class S> __device__ __forceinline__ bool park(const RayState& r, const S& st) {
        if (last) return false;
#ifdef IRIS_TAIL_NEVER
        if (r.h.t < -1e30f) last = true; else return false;
#endif
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const bool act = r.cur != kEmptyRef;
        const unsigned long long m = __ballot(act);
        if (act) {                                // (speculative: without the bit in `state` nobody looks at these)
            const int i = wave * kTailMax + __popcll(m & ((1ull << lane) - 1ull));
            t->hit[i] = make_float4(r.h.t, r.h.u, r.h.v, __int_as_float(r.h.slot));
            t->cur[i] = r.cur;
            t->id[i] = r.h.id;
            t->meta[i] = (uint32_t)(my_r & 0xffff) | (uint32_t)st.sp << 16 | (uint32_t)threadIdx.x << 24;
        }
        int ok = 0;
        if (lane == 0) {
            t->count[wave] = (uint32_t)__popcll(m);
            uint32_t s = __hip_atomic_load(&t->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while ((s & 0xffu) >= 2u) {
                if (__hip_atomic_compare_exchange_strong(&t->state, &s, (s - 1u) | (1u << (8 + wave)), __ATOMIC_RELEASE, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP)) { ok = 1; break; }
            }
        }
        ok = __shfl(ok, 0);
        if (!ok) last = true;
        return ok != 0;
    }
    // whole wave, no rays: true = gone
    __device__ __forceinline__ bool leave() {
        int ok = 0;
        if ((threadIdx.x & 63) == 0) {
            uint32_t s = load_state();
            for (;;) {
                if ((s & 0xffu) <= 1u && unclaimed(s)) break;        // the last one: adopt them first
                if (__hip_atomic_compare_exchange_strong(&t->state, &s, s - 1u, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) { ok = 1; break; }
            }
        }
        return __shfl(ok, 0) != 0;
    }
};

template <int LAYOUT, bool COUNT, int CAP, int TILE_STACK, bool GLOBAL_OVF, bool HANDOVER, class PhaseA, class FetchRay, class Prepare, class Retire>
__device__ __forceinline__ void tile_sort_trace(const SceneDev& sc, int nr, uint16_t* s_sorted, uint32_t* s_stack, int* s_chunk, uint32_t* ovf,
                                                TileHandOver* hand, TraceStats& ts, PhaseA phase_a, FetchRay fetch_ray, Prepare prepare, Retire retire) {
    static_assert(TILE_STACK * kBlock * 4 >= CAP + 2 * 256 * 4, "stack region too small to alias the sort keys");
    uint8_t* s_keys = reinterpret_cast<uint8_t*>(s_stack);
    uint32_t* s_hist = s_stack + CAP / 4;
    uint32_t* s_cur = s_hist + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    IRIS_PHASE_BEGIN();
    static_assert(!HANDOVER || GLOBAL_OVF, "the hand-over copies stack columns: LDS + the workgroup's overflow slab only");
    if (HANDOVER) {        // (published by the barriers below, long before the first wave can park)
        if (tid < 4) { hand->count[tid] = 0; hand->taken[tid] = 0; }
        if (tid == 0) hand->state = kBlock / 64;
    }
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
        if constexpr (HANDOVER) {
            TileTailOps<TILE_STACK, FetchRay> tail{hand, my_r, fetch_ray, s_stack, ovf, false};
            trace_stream<LAYOUT, COUNT, TILE_STACK, GLOBAL_OVF>(sc, s_stack + tid, ovf, &ts, fetch, prepare, ret, tail);
        } else {
            trace_stream<LAYOUT, COUNT, TILE_STACK, GLOBAL_OVF>(sc, s_stack + tid, ovf, &ts, fetch, prepare, ret);
        }
        IRIS_PHASE_MARK(4);      // wave 0's own traversal; 2 (below) also counts its wait for the slowest wave of the tile
    }
    __syncthreads();
    IRIS_PHASE_MARK(2);
}

}  // namespace iris
