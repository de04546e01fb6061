// BVH traversal + watertight ray / triangle closest hit (replaces utils/path_tracing.py:30-43, the Mitsuba/OptiX call).
// One ray per lane, per-lane traversal stack in LDS (stack[k*BLOCK + tid] -> conflict-free for ds_read/write_b32).
#pragma once
#include "iris_device.h"

namespace iris {

constexpr int kBlock = 256;          // threads per workgroup for every traversal kernel
constexpr int kStackLds = 24;        // per-lane stack entries kept in LDS (24 KiB per workgroup)
constexpr int kStackSpill = 72;      // rarely-touched overflow in scratch (correctness only)
// Q8 node record.  IRIS_NODE80 = 0: 64 B, 8-bit planes four to a word (a visit isolates near / far byte pairs with 12 v_perm_b32).
// IRIS_NODE80 = 1 (round 5): 80 B = {origin.xyz, scale} {ref[4]} {12 plane words}: every (axis, child) has its OWN word, near byte in the low half, far byte in the
// high half, each zero-extended to 16 bits -- i.e. ALREADY the pair of f16 subnormals v_fma_mix_f32 reads: no v_perm_b32 in a visit (12 of its 91 vector
// instructions, all of the 4-cycle class), five 16-B loads instead of four, ONE plane scale per node (the largest axis extent; 80 B = five loads exactly) --
// tools/bvh_eval: +0.8 % node visits, +1.6 % triangle tests from the coarser planes on a node's short axes.
#ifndef IRIS_NODE80
#define IRIS_NODE80 0
#endif
#ifndef IRIS_SLAB_CVT
#define IRIS_SLAB_CVT 0
#endif
#ifndef IRIS_EXP_NODRAIN
#define IRIS_EXP_NODRAIN 0
#endif
constexpr uint32_t kNodeBytes = IRIS_NODE80 ? 80u : 64u;
__device__ __forceinline__ uint32_t node_offset(uint32_t cur) { return IRIS_NODE80 ? cur * 80u : cur << 6; }
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kEmptyRef = 0xFFFFFFFFu;

// Scene as laid out in HBM
struct SceneDev {
    const float4* nodes;  // layout-dependent; BVH4_F32: 8 x float4 = 128 B per node, 128-B aligned
    const float4* tris;   // 64 B per leaf triangle (one L2 half-line), component-major: (p0.x,p1.x,p2.x,id) (p0.y,p1.y,p2.y,id) (p0.z,p1.z,p2.z,id) (0,0,0,0)
    int n_nodes;
    int n_tris;
    int phase_min;  // wave-level phase scheduling threshold (see trace_bvh4)
    int layout;     // kLayoutF32 | kLayoutQ8
    uint32_t oct_stride;   // Q8: bytes between the node-table copies of two ray octants (n_nodes * 64)
};

struct Hit {
    float t, u, v;
    int slot;  // index into SceneDev::tris, -1 = miss
    int id;    // original triangle index
};

__device__ __forceinline__ float safe_rcp_dir(float d) {
    // keeps (box - o) * idir finite for axis-parallel rays
    // v_rcp_f32 (1 ulp) instead of the 11-instruction IEEE division: the slab test only has to be conservative, and the boxes are padded by
    // 2e-5 of the scene extent against 6e-8 relative here; hits are computed from o and d, never from this
    return fabsf(d) < 1e-20f ? copysignf(1e20f, d) : __builtin_amdgcn_rcpf(d);    // (the quantised nodes store scale * 2^24: scale * idir must stay finite too)
}

// Per-lane stack: the first LDS_DEPTH entries in LDS; deeper entries (a few % of the rays at depth 10-12) either in a private
// (scratch) array or -- GLOBAL_OVF, kernels that must not use scratch -- in a workgroup-private slab of the caller's workspace
// laid out [entry][thread] so that the rare accesses coalesce.
constexpr int kStackCapacity = kStackLds + kStackSpill;
// The LDS part is addressed through an address-space-3 pointer: with a generic pointer hipcc merges the LDS and the overflow path of
// pop() into one flat_load_dword (a vector-memory instruction with an aperture check, counted on vmcnt AND lgkmcnt) -- on the dependent
// chain of every pop; with the qualified pointer it is a ds_read_b32 and the overflow a global_load behind a (rare) branch.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// (the table reads are addressed through address-space-1 pointers for the same reason: a table base that went through an optimisation barrier
//  -- tile_sort_trace -- is no longer known to be global, and hipcc falls back to flat_load)
//  (native vector types: HIP's float4 / uint4 are classes, and copying one out of an address-space-1 lvalue goes through a generic reference again)
typedef uint32_t iris_u4v __attribute__((ext_vector_type(4)));
typedef float iris_f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const iris_u4v glb_u4v;
typedef __attribute__((address_space(1))) const iris_f4v glb_f4v;
template <int LDS_DEPTH, bool GLOBAL_OVF = false>
struct Stack {
    lds_u32* lds;   // &s_stack[threadIdx.x]
    uint32_t* ovf;  // GLOBAL_OVF: the workgroup's slab (wave-uniform: stays in scalar registers; the lane offset is added at the rare use)
    uint32_t tid;   // threadIdx.x as the caller derived it for THIS traversal (tile kernels: a fresh value per phase, see tile_sort_trace)
    uint32_t spill[GLOBAL_OVF ? 1 : kStackCapacity - LDS_DEPTH];
    int sp;
    __device__ __forceinline__ void push(uint32_t v) {
        if (sp < LDS_DEPTH) lds[sp * kBlock] = v;
        else if (GLOBAL_OVF) ovf[(uint32_t)min(sp - LDS_DEPTH, kStackCapacity - LDS_DEPTH - 1) * kBlock + tid] = v;
        else spill[min(sp - LDS_DEPTH, kStackCapacity - LDS_DEPTH - 1)] = v;
        ++sp;
    }
    // Three pushes without a branch (the caller has checked, for the whole wave, that sp <= LDS_DEPTH - 3): every reference is WRITTEN at the current top and the
    // top moves only if its condition holds (t >= 0) -- a reference that is not pushed leaves garbage in a free slot.  No execution masks, no overflow tests.
    __device__ __forceinline__ void push3_fast(uint32_t v3, int32_t t3, uint32_t v2, int32_t t2, uint32_t v1, int32_t t1) {
        int u = sp;
        lds[u * kBlock] = v3; u += 1 + (t3 >> 31);
        lds[u * kBlock] = v2; u += 1 + (t2 >> 31);
        lds[u * kBlock] = v1; u += 1 + (t1 >> 31);
        sp = u;
    }
    __device__ __forceinline__ uint32_t pop_lds() { --sp; return lds[sp * kBlock]; }   // (the caller knows that the top is in LDS)
    __device__ __forceinline__ uint32_t pop() {
        --sp;
        if (sp < LDS_DEPTH) return lds[sp * kBlock];
        if (GLOBAL_OVF) return ovf[(uint32_t)min(sp - LDS_DEPTH, kStackCapacity - LDS_DEPTH - 1) * kBlock + tid];
        return spill[min(sp - LDS_DEPTH, kStackCapacity - LDS_DEPTH - 1)];
    }
};

#ifndef IRIS_TRI_FLAT
#define IRIS_TRI_FLAT 1
#endif
// Watertight ray / triangle test (Woop, Benthin, Wald 2013, "Watertight Ray/Triangle Intersection", JCGT 2(1), section 3): the ray is
// made the z axis of a sheared, axis-permuted space -- kz = the axis of d's largest magnitude, (kx, ky) the two after it, S = (d[kx], d[ky], 1) / d[kz]
// --, every vertex is translated by the origin and sheared by the same function of (vertex, ray) in whichever triangle it appears, and the
// three 2-D edge functions are evaluated with plain products and a plain difference: rnd(a*b) - rnd(c*d) is exactly antisymmetric in the two
// vertices (the two triangles sharing an edge see opposite values) and never has the opposite sign of the exact value; a value of 0 is
// re-evaluated in double (exact sign).  So no ray can slip between two triangles that share an edge or a vertex.
// The axis permutation costs nothing here: a leaf record stores the triangle component-major -- (p0.x,p1.x,p2.x,id) (p0.y,..) (p0.z,..) -- and a
// lane reads plane kx / ky / kz by ADDRESS (three 16-B loads from one 64-B line either way).  Arithmetic contract of oracle/iris_oracle.c.
struct RayXf {
    float ox, oy, oz;   // origin, permuted: o[kx], o[ky], o[kz]
    float sx, sy, sz;   // shear: d[kx] * sz, d[ky] * sz, sz = 1 / d[kz]
    uint32_t offx, offy, offz;   // byte offsets of planes kx, ky, kz inside a leaf record
};
__device__ __forceinline__ uint32_t ray_kz(f3 d) {
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    return (ax >= ay && ax >= az) ? 0u : (ay >= az ? 1u : 2u);
}
__device__ __forceinline__ float pick3(f3 v, uint32_t k) { return k == 0u ? v.x : (k == 1u ? v.y : v.z); }
__device__ __forceinline__ void ray_xform(f3 o, f3 d, RayXf& x) {
    const uint32_t kz = ray_kz(d), kx = kz == 2u ? 0u : kz + 1u, ky = kx == 2u ? 0u : kx + 1u;
    x.sz = 1.0f / pick3(d, kz);
    x.sx = pick3(d, kx) * x.sz; x.sy = pick3(d, ky) * x.sz;
    x.ox = pick3(o, kx); x.oy = pick3(o, ky); x.oz = pick3(o, kz);
    x.offx = kx << 4; x.offy = ky << 4; x.offz = kz << 4;
}

// The test on leaf record `slot`.  Accept iff the edge functions have no two strictly opposite signs, det = U + V + W != 0 and 0 <= t < inf;
// (b1, b2) = (V, W) / det (p = b0 p0 + b1 p1 + b2 p2), t = (U Az + V Bz + W Cz) * sz / det (z = P[kz] - o[kz]).
__device__ __forceinline__ void tri_load(const SceneDev& sc, int slot, const RayXf& x, iris_f4v& X, iris_f4v& Y, iris_f4v& Z) {
    const uint32_t base = (uint32_t)slot << 6;     // 32-bit byte offset from the (scalar) table base
    const char* tb = reinterpret_cast<const char*>(sc.tris);
    X = *(glb_f4v*)(tb + (size_t)(base + x.offx));
    Y = *(glb_f4v*)(tb + (size_t)(base + x.offy));
    Z = *(glb_f4v*)(tb + (size_t)(base + x.offz));
}
__device__ __forceinline__ void tri_eval(const iris_f4v X, const iris_f4v Y, const iris_f4v Z, int slot, const RayXf& x, Hit& h) {
    const int id = __float_as_int(Z.w);            // (every plane carries the index)
    const float Atz = Z.x - x.oz, Btz = Z.y - x.oz, Ctz = Z.z - x.oz;
    const float Ax = fmaf(-x.sx, Atz, X.x - x.ox), Ay = fmaf(-x.sy, Atz, Y.x - x.oy);
    const float Bx = fmaf(-x.sx, Btz, X.y - x.ox), By = fmaf(-x.sy, Btz, Y.y - x.oy);
    const float Cx = fmaf(-x.sx, Ctz, X.z - x.ox), Cy = fmaf(-x.sy, Ctz, Y.z - x.oy);
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (fminf(fminf(fabsf(U), fabsf(V)), fabsf(W)) == 0.f) {      // any of them 0 (rare: the ray through an edge or a vertex of the projected triangle, or an underflow)
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    const float det = (U + V) + W;
    const float inv_det = 1.0f / det;
    const float t = (fmaf(U, Atz, fmaf(V, Btz, W * Ctz)) * x.sz) * inv_det;
    const float u = V * inv_det, v = W * inv_det;
    // no two edge functions of strictly opposite sign; det == 0 gives t = NaN or +-inf: a NaN fails every comparison, and +inf can only tie with the
    // initial h.t, whose h.id = INT_MIN no index is smaller than
#if IRIS_TRI_FLAT
    // (bitwise, not short-circuit: as `&&` / `||` hipcc turns the acceptance test into four divergent branches with their execution-mask bookkeeping -- ~25 scalar
    //  instructions per triangle test; the same comparisons combined as masks are five compares and four scalar mask operations)
    const bool mixed = (int)(fminf(fminf(U, V), W) < 0.f) & (int)(fmaxf(fmaxf(U, V), W) > 0.f);
    const bool ok = (int)!mixed & (int)(t >= 0.f);
    const bool closer = (int)(t < h.t) | ((int)(t == h.t) & (int)(id < h.id));       // closest hit = lexicographic min of (t, original index)
    if ((int)ok & (int)closer) { h.t = t; h.u = u; h.v = v; h.slot = slot; h.id = id; }
#else
    const bool ok = !(fminf(fminf(U, V), W) < 0.f && fmaxf(fmaxf(U, V), W) > 0.f) && t >= 0.f;
    // closest hit = lexicographic min of (t, original index)
    if (ok && (t < h.t || (t == h.t && id < h.id))) { h.t = t; h.u = u; h.v = v; h.slot = slot; h.id = id; }
#endif
}

__device__ __forceinline__ void tri_test(const SceneDev& sc, int slot, const RayXf& x, Hit& h) {
    iris_f4v X, Y, Z;
    tri_load(sc, slot, x, X, Y, Z);
    tri_eval(X, Y, Z, slot, x, h);
}

// -------------------------------------------------------------------------------------------------------
// BVH4_F32 traversal: node = {lox[4],hix[4],loy[4],hiy[4],loz[4],hiz[4],ref[4],pad[4]} (128 B, one L2 line).
// ref: internal -> node index; leaf -> 0x80000000 | start<<3 | count; unused slot -> inverted box (never hit).
// -------------------------------------------------------------------------------------------------------
#define IRIS_CE(ka, ra, kb, rb) { bool sw = kb < ka; float tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; \
                                  uint32_t tr = sw ? rb : ra; rb = sw ? ra : rb; ra = tr; }
#define IRIS_HITKEY(tn, tf) ((tn) <= (tf) ? (tn) : INFINITY)

// (Loop control: `__popcll(m) >= k` stays a 64-bit comparison, which the scalar ALU cannot do, so hipcc runs it on the vector ALU -- v_cmp_lt_u64 on scalar
//  operands, in front of every node and leaf step.  Counting the two mask halves with 32-bit scalar instructions instead was measured SLOWER, -0.8 %: the longer
//  scalar dependency chain in front of the step costs more than the vector issue slot, EXPERIMENTS.md round 4.)
#ifndef IRIS_POPC_ASM
#define IRIS_POPC_ASM 1
#endif
#ifndef IRIS_IDLE_GATE
#define IRIS_IDLE_GATE 1
#endif
#ifndef IRIS_LOOP_NEST
#define IRIS_LOOP_NEST 1
#endif
#ifndef IRIS_FAST_PUSH
#define IRIS_FAST_PUSH 1
#endif
#ifndef IRIS_FAST_POP
#define IRIS_FAST_POP 1
#endif
__device__ __forceinline__ int popc_mask(unsigned long long m) {
#if IRIS_POPC_ASM
    int n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");     // ONE scalar instruction and a 32-bit result: the comparison that follows stays on the scalar ALU
    return n;
#else
    return (int)__popcll(m);
#endif
}
__device__ __forceinline__ bool first_active_lane() {
    unsigned long long m = __ballot(1);
    return (int)(threadIdx.x & 63) == (__ffsll((long long)m) - 1);
}
// Traversal statistics (instrumented builds only): per-lane counts, reduced by the caller.
struct TraceStats {
    uint32_t nodes = 0;       // node visits of this lane
    uint32_t tris = 0;        // triangle tests of this lane
    uint32_t node_iters = 0;  // wave-level executions of the node step (counted by the first active lane)
    uint32_t leaf_iters = 0;  // wave-level executions of the triangle test
    uint32_t sp_gt8 = 0, sp_gt12 = 0, sp_gt16 = 0;  // rays whose stack ever exceeded 8 / 12 / 16 entries
    unsigned long long max_steps64 = 0;             // sum over wave-chunks of 64 * (longest ray of the chunk, in node+triangle steps)
    uint32_t drain_nodes = 0, drain_node_iters = 0; // trace_stream: the same two node counters while the ray list is exhausted (no refill)
    uint32_t top21 = 0, top85 = 0, top341 = 0, top1365 = 0;   // node visits with node index < 21 / 85 / 341 / 1365 (nodes are in breadth-first order:
                                                              // the first 1 + 4 + 16 (+ 64 (+ 256 (+ 1024))) nodes are the top 3 (4, 5, 6) levels of a full tree)
    uint32_t slow_push_iters = 0;     // wave node iterations that could not take the branch-free pushes (some lane's stack within three entries of the LDS part's end)
    uint32_t shared_path_iters = 0;   // wave node iterations EXECUTED through the scalar path (node_step_shared: >= IRIS_SCALAR_TOP lanes at one node, test not gated off)
    uint32_t shared_iters = 0, shared_lanes = 0, shared_all_iters = 0;   // wave node iterations in which >= 32 of the lanes at a node sit at the SAME node of the same
                                                                          // octant table (counted by the first active lane), the lanes that share it, and the
                                                                          // iterations in which every lane at a node does
    __device__ __forceinline__ void count_shared(uint32_t cur, uint32_t oct_base) {   // called by the lanes at a node
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur), o0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)oct_base);
        const int n_same = __popcll(__ballot(cur == c0 && oct_base == o0)), n_all = __popcll(__ballot(1));
        if (first_active_lane() && n_same >= 32) { shared_iters++; shared_lanes += (uint32_t)n_same; shared_all_iters += n_same == n_all; }
    }
    __device__ __forceinline__ void count_top(uint32_t cur) { top21 += cur < 21u; top85 += cur < 85u; top341 += cur < 341u; top1365 += cur < 1365u; }
};

// Wave-level phase scheduling.  A lane is in one of three states: at an internal node, at a leaf (k triangles tested so far),
// or done.  The wave alternates a NODE phase and a LEAF phase; a phase ends when no lane needs it, or early when fewer than
// kPhaseMin lanes still need it while at least kPhaseMin lanes wait for the other phase (stragglers are carried over instead of
// keeping the whole wave in a nearly empty phase).  Per-lane results do not depend on the schedule.
constexpr int kPhaseMin = 12;      // (round 3 sweep at 4096-ray tiles: 10: 7.47, 12: 7.50, 14: 7.50, 16: 7.47, 20: 7.32 Grays/s)

// Node layouts (iris_hip.h): BVH4_F32 = 128-B node with f32 planes (7 dwordx4 per visit); BVH4_Q8 = 64-B node
// {origin.xyz, scale.x | scale.y, scale.z, qlo_x, qlo_y | qlo_z, qhi_x, qhi_y, qhi_z | ref[4]} with 8-bit planes relative to the node's
// own box (4 dwordx4 per visit): plane = origin + q * 2^e per axis (the node stores 2^(e+24) as a float, see node_step), lo rounded down /
// hi rounded up, so the decoded box contains the f32 box.
constexpr int kLayoutF32 = 1, kLayoutQ8 = 3;
__device__ __forceinline__ float ubyte(uint32_t v, int c) { return (float)((v >> (8 * c)) & 0xffu); }

// Per-lane traversal state shared by the two drivers below (kept in registers; the struct is scalar-replaced).
struct RayState {
    f3 o, d;                        // (scratch between fetch and prepare; the traversal itself reads the slab constants and the triangle test's transform)
    float tox, toy, toz, sx, sy, sz;   // triangle test: permuted origin, shear (RayXf)
    uint32_t kz;                    // its axis permutation; the three plane offsets are rebuilt from it at every leaf-phase entry
    uint32_t oct_base;              // Q8 nodes: byte offset of the node-table copy of this ray's octant
    float ix, iy, iz, nx, ny, nz;   // 1/d and -o/d
    bool px, py, pz;                // direction signs
    Hit h;
    uint32_t cur;                   // node / leaf reference (a leaf reference is consumed in place: start + 1, count - 1 per test), kEmptyRef = no work
};
__device__ __forceinline__ void ray_begin(const SceneDev& sc, RayState& r, f3 o, f3 d) {
    r.o = o; r.d = d;
    r.h.t = INFINITY; r.h.u = 0.f; r.h.v = 0.f; r.h.slot = -1; r.h.id = (int)0x80000000;   // (INT_MIN: see tri_test)
    r.ix = safe_rcp_dir(d.x); r.iy = safe_rcp_dir(d.y); r.iz = safe_rcp_dir(d.z);
    r.nx = -(o.x * r.ix); r.ny = -(o.y * r.iy); r.nz = -(o.z * r.iz);
    r.px = r.ix >= 0.f; r.py = r.iy >= 0.f; r.pz = r.iz >= 0.f;
    r.oct_base = ((r.px ? 0u : 1u) | (r.py ? 0u : 2u) | (r.pz ? 0u : 4u)) * sc.oct_stride;
    { RayXf x; ray_xform(o, d, x); r.tox = x.ox; r.toy = x.oy; r.toz = x.oz; r.sx = x.sx; r.sy = x.sy; r.sz = x.sz; r.kz = x.offz; }
    r.cur = 0;
}

// One internal-node visit of the lanes that are at a node: 4 slab tests, 5-comparator sorting network, near child first.
template <int LAYOUT, class STACK>
__device__ __forceinline__ void node_step(const SceneDev& sc, RayState& r, STACK& st, bool fast_push = false) {
    float k0, k1, k2, k3;
    uint32_t r0, r1, r2, r3;
    const float ix = r.ix, iy = r.iy, iz = r.iz, nx = r.nx, ny = r.ny, nz = r.nz;
    const bool px = r.px, py = r.py, pz = r.pz;
    if (LAYOUT == kLayoutQ8) {
        // The node table exists once per ray octant (iris_hip.hip): the copy a ray reads holds the children in ITS front-to-back order (the order of
        // the binary splits the node was collapsed from) and, per axis, the plane it meets first in the "near" bytes -- so a visit selects no planes
        // by the ray's signs and sorts nothing.  32-bit byte offset from the (scalar) table base: one shift-add per visit.
        glb_u4v* n = (glb_u4v*)(reinterpret_cast<const char*>(sc.nodes) + (size_t)(uint32_t)(node_offset(r.cur) + r.oct_base));
        typedef _Float16 iris_h2 __attribute__((ext_vector_type(2)));
#if IRIS_NODE80
        iris_u4v hd = n[0], rf = n[1], px = n[2], py = n[3], pz = n[4];            // {origin.xyz, scale} {ref[4]} {x planes of children 0..3} {y planes} {z planes}
        // (all five 16-B loads HERE: left alone hipcc narrows the plane loads to single words and sinks those of children 1..3 behind the test of child 0 --
        //  dependent round trips inside a visit)
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
        r0 = rf.x; r1 = rf.y; r2 = rf.z; r3 = rf.w;
        const float sc24 = __uint_as_float(hd.w);                                   // 2^(e+24), one per node
        const float ax = sc24 * ix, ay = sc24 * iy, az = sc24 * iz;
        const float bx = fmaf(__uint_as_float(hd.x), ix, nx), by = fmaf(__uint_as_float(hd.y), iy, ny), bz = fmaf(__uint_as_float(hd.z), iz, nz);
        // a plane word IS the (near, far) pair of f16 subnormals q * 2^-24: v_fma_mix_f32 reads its halves directly
#define IRIS_SLABQ(D, C)                                                                                                          \
    {                                                                                                                             \
        const iris_h2 hx = __builtin_bit_cast(iris_h2, (uint32_t)px[C]), hy = __builtin_bit_cast(iris_h2, (uint32_t)py[C]), hz = __builtin_bit_cast(iris_h2, (uint32_t)pz[C]);   /* (a prvalue: __builtin_bit_cast of the vector ELEMENT lvalue reads element 0 whatever the index -- hipcc 7.2) */ \
        float tn = fmaxf(fmaxf(fmaf((float)hx.x, ax, bx), fmaf((float)hy.x, ay, by)), fmaxf(fmaf((float)hz.x, az, bz), 0.f));      \
        float tf = fminf(fminf(fmaf((float)hx.y, ax, bx), fmaf((float)hy.y, ay, by)), fminf(fmaf((float)hz.y, az, bz), r.h.t));    \
        D = tf - tn;                                                                                                              \
    }
#else
        const iris_u4v hd = n[0], q1 = n[1], q2 = n[2], rf = n[3];
        r0 = rf.x; r1 = rf.y; r2 = rf.z; r3 = rf.w;
        // per-axis: t(q) = q * 2^e * idir + (origin * idir - o * idir); the node stores 2^(e+24) as a float (see below)
        const float ax = __uint_as_float(hd.w) * ix, ay = __uint_as_float(q1.x) * iy, az = __uint_as_float(q1.y) * iz;
        const float bx = fmaf(__uint_as_float(hd.x), ix, nx), by = fmaf(__uint_as_float(hd.y), iy, ny), bz = fmaf(__uint_as_float(hd.z), iz, nz);
        const uint32_t nxq = q1.z, nyq = q1.w, nzq = q2.x, fxq = q2.y, fyq = q2.z, fzq = q2.w;
        // A plane byte q, zero-extended to 16 bits, IS the f16 subnormal q * 2^-24; v_perm_b32 puts the near and the far byte of one
        // child into the two halves of a register and v_fma_mix_f32 reads an f16 operand directly: 1 + 2 instructions per axis and child
        // instead of 2 conversions + 2 FMAs.  The node stores scale * 2^24, so q*2^-24 * (scale*2^24*idir) + b is the same real number,
        // rounded once by the FMA.
#if IRIS_SLAB_CVT
        // (round 5 experiment) the plane bytes converted by v_cvt_f32_ubyteN and fed to PLAIN v_fma_f32: 12 instructions per child instead of 9, but of two classes that
        // issue TOGETHER (tools/microbench pair: an integer-pipe instruction followed by a float-pipe one costs ~5 cycles the pair; v_fma_mix_f32 pairs with nothing, 4.4
        // cycles each).  q * (scale * idir) + b is the same real number as (q * 2^-24) * (scale * 2^24 * idir) + b: the same t, bit for bit.
        const float ax1 = ax * 0x1p-24f, ay1 = ay * 0x1p-24f, az1 = az * 0x1p-24f;
#define IRIS_SLABQ(D, C)                                                                                                          \
    {                                                                                                                             \
        float tn = fmaxf(fmaxf(fmaf(ubyte(nxq, C), ax1, bx), fmaf(ubyte(nyq, C), ay1, by)), fmaxf(fmaf(ubyte(nzq, C), az1, bz), 0.f));      \
        float tf = fminf(fminf(fmaf(ubyte(fxq, C), ax1, bx), fmaf(ubyte(fyq, C), ay1, by)), fminf(fmaf(ubyte(fzq, C), az1, bz), r.h.t));    \
        D = tf - tn;                                                                                                              \
    }
#define IRIS_PLANES(NQ, FQ, C) 0
#else
#define IRIS_PLANES(NQ, FQ, C) __builtin_bit_cast(iris_h2, __builtin_amdgcn_perm(NQ, FQ, 0x0c000c04u | ((uint32_t)(C) << 16) | (uint32_t)(C)))
#define IRIS_SLABQ(D, C)                                                                                                          \
    {                                                                                                                             \
        const iris_h2 hx = IRIS_PLANES(nxq, fxq, C), hy = IRIS_PLANES(nyq, fyq, C), hz = IRIS_PLANES(nzq, fzq, C);                 \
        float tn = fmaxf(fmaxf(fmaf((float)hx.x, ax, bx), fmaf((float)hy.x, ay, by)), fmaxf(fmaf((float)hz.x, az, bz), 0.f));      \
        float tf = fminf(fminf(fmaf((float)hx.y, ax, bx), fmaf((float)hy.y, ay, by)), fminf(fmaf((float)hz.y, az, bz), r.h.t));    \
        D = tf - tn;                     /* sign clear: the child is hit (tn <= tf).  tn = tf gives +0; a NaN (inf - inf: tn = tf = inf) with a clear sign  */ \
    }                                    /* would only cost a wasted visit                                                                                */
#endif
#endif
        float d0, d1, d2, d3;
        IRIS_SLABQ(d0, 0) IRIS_SLABQ(d1, 1) IRIS_SLABQ(d2, 2) IRIS_SLABQ(d3, 3)
#undef IRIS_SLABQ
#if !IRIS_NODE80
#undef IRIS_PLANES
#endif
        // Slots are in visiting order: the first child hit is next, the others wait on the stack, the farthest at the bottom.  The conditions are
        // taken from the SIGN BITS of the interval lengths with integer and / or (2-cycle dual-issue instructions; boolean algebra on compare results is
        // what hipcc turns into 0 / 1 registers, and fminf / fmaxf bring a canonicalising v_max x, x per operand): "child j is hit" = sign clear.
        const int32_t b0 = __float_as_int(d0), b1 = __float_as_int(d1), b2 = __float_as_int(d2), b3 = __float_as_int(d3);
        const int32_t n01 = b0 & b1, n012 = n01 & b2;           // sign set: none of these children is hit
        // (the references are needed whichever child is hit: this keeps their load with the plane loads instead of behind the hit test, where hipcc
        //  sinks it otherwise -- a second dependent round trip per visit, measured -8 %)
        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        if (fast_push) {                                        // (wave-uniform)
            uint32_t c = b2 >= 0 ? r2 : r3;
            c = b1 >= 0 ? r1 : c;
            c = b0 >= 0 ? r0 : c;
            st.push3_fast(r3, b3 | n012, r2, b2 | n01, r1, b1 | b0);      // (no child hit: every condition's sign is set, nothing moves)
            if ((n012 & b3) >= 0) r.cur = c;
            else r.cur = st.sp > 0 ? st.pop_lds() : kEmptyRef;       // (a branch-free pop -- every lane reads the entry below its top, selects afterwards -- was measured -2 %:
            return;                                                  //  it puts an LDS round trip behind EVERY node step)
        }
        if ((n012 & b3) >= 0) {
            uint32_t c = b2 >= 0 ? r2 : r3;                     // (selects, written innermost first: as a nested ?: hipcc makes this three branches)
            c = b1 >= 0 ? r1 : c;
            r.cur = b0 >= 0 ? r0 : c;
            if ((b3 | n012) >= 0) st.push(r3);                  // hit, and a nearer child is hit too
            if ((b2 | n01) >= 0) st.push(r2);
            if ((b1 | b0) >= 0) st.push(r1);
        } else {
            r.cur = st.sp > 0 ? st.pop() : kEmptyRef;
        }
        return;
    } else {
        const float4* n = sc.nodes + (int64_t)r.cur * 8;
        const float4 lox = n[0], hix = n[1], loy = n[2], hiy = n[3], loz = n[4], hiz = n[5];
        const float4 rf = n[6];
        r0 = __float_as_uint(rf.x); r1 = __float_as_uint(rf.y); r2 = __float_as_uint(rf.z); r3 = __float_as_uint(rf.w);
#define IRIS_SLAB(K, C)                                                                                               \
    {                                                                                                                 \
        float tn = fmaxf(fmaxf(fmaf(px ? lox.C : hix.C, ix, nx), fmaf(py ? loy.C : hiy.C, iy, ny)),                   \
                         fmaxf(fmaf(pz ? loz.C : hiz.C, iz, nz), 0.f));                                               \
        float tf = fminf(fminf(fmaf(px ? hix.C : lox.C, ix, nx), fmaf(py ? hiy.C : loy.C, iy, ny)),                   \
                         fminf(fmaf(pz ? hiz.C : loz.C, iz, nz), r.h.t));                                             \
        K = tn <= tf ? tn : INFINITY;                                                                                 \
    }
        IRIS_SLAB(k0, x) IRIS_SLAB(k1, y) IRIS_SLAB(k2, z) IRIS_SLAB(k3, w)
#undef IRIS_SLAB
    }
    IRIS_CE(k0, r0, k1, r1) IRIS_CE(k2, r2, k3, r3) IRIS_CE(k0, r0, k2, r2) IRIS_CE(k1, r1, k3, r3) IRIS_CE(k1, r1, k2, r2)
    if (k0 < INFINITY) {
        r.cur = r0;
        if (k3 < INFINITY) st.push(r3);
        if (k2 < INFINITY) st.push(r2);
        if (k1 < INFINITY) st.push(r1);
    } else {
        r.cur = st.sp > 0 ? st.pop() : kEmptyRef;
    }
}
// A node visit made TOGETHER (round 4).  Instrumented launches say that in 36 % of the node steps at least 32 of the lanes at a node sit at the SAME node
// of the same octant table -- rays of one direction bin, refilled together, walk the top of the tree together: 48 % of all node visits, 51 lanes at a time.
// When at least IRIS_SCALAR_TOP lanes share the node (byte offset off0, wave-uniform), its 16 words come through the SCALAR cache (one s_load_dwordx16), the
// plane bytes are isolated on the scalar ALU (a byte zero-extended in a scalar register IS the f16 subnormal v_fma_mix_f32 reads) and the slab tests take
// them as scalar operands: no vector loads, no address arithmetic, no v_perm_b32 for that visit (50 instead of 63 vector instructions in front of the
// push logic).  Same arithmetic, same bits; the few lanes at other nodes sit the step out.  Measured +2.5 % at a threshold of 44 lanes (32: +0.7 %,
// 38-47: +1.0 ... +1.2 % before the detection reused the loop head's ballot, 50: -0.4 %, 56: -2.2 %: below ~36 lanes the lanes sitting out cost more than
// the shared visit saves, above ~48 the test itself -- two vector instructions in front of EVERY node step -- is paid too often for nothing).  0 = off.
// The test costs two vector instructions and a scalar chain in front of a node step, and deep in the tree it never hits: after IRIS_SHARED_TRIES misses in a
// row a wave skips it until its next refill (+1.3 ... +1.6 % on top: 1, 2, 3 misses equal, 4: +1.2 %, 8: +0.4 %; taking the reference lane from the rays of the
// latest refill instead of the first lane at a node: no difference; a threshold of 36 / 40 with it: -0.4 / -0.6 %).
#ifndef IRIS_SCALAR_TOP
#define IRIS_SCALAR_TOP 44
#endif
#ifndef IRIS_SHARED_TRIES
#define IRIS_SHARED_TRIES 2
#endif
constexpr int kSharedTries = IRIS_SHARED_TRIES;
#ifndef IRIS_SHARED_PAIRS
#define IRIS_SHARED_PAIRS 1
#endif
#if IRIS_SCALAR_TOP
typedef __attribute__((address_space(4))) const uint32_t cst_u32;
template <class STACK>
__device__ __forceinline__ void node_step_shared(const SceneDev& sc, RayState& r, STACK& st, uint32_t off0, bool fast_push = false) {
    // (inline asm: through a pointer hipcc proves the table global, falls back to four vector loads of the uniform address and turns the byte pairing
    //  below back into v_perm_b32.  Issuing the load BEFORE the test, so that its latency runs under the compare / count / branch -- with a wait in the
    //  path not taken, whose registers must not be reused while it is in flight --: no gain, -0.2 %.)
    typedef uint32_t iris_u16v __attribute__((ext_vector_type(16)));
    typedef _Float16 iris_h2 __attribute__((ext_vector_type(2)));
    const uint64_t base = reinterpret_cast<uint64_t>(sc.nodes);
    const uint32_t off_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)off0);    // (inside `if (off == off0)` hipcc substitutes the per-lane value for the uniform one)
    const float ix = r.ix, iy = r.iy, iz = r.iz, nx = r.nx, ny = r.ny, nz = r.nz;
#if IRIS_NODE80
    // 20 words: {origin.xyz, scale} {ref[4]} {x planes of children 0..3} {y planes} | {z planes}: one x16 and one x4 scalar load; a plane word is the (near, far)
    // pair of f16 subnormals as it stands -- no byte isolation on the scalar ALU either
    typedef uint32_t iris_u4s __attribute__((ext_vector_type(4)));
    iris_u16v w; iris_u4s wz;
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %3 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w), "=&s"(wz) : "s"(base), "s"(off_s) : "memory");
    uint32_t r0 = w[4], r1 = w[5], r2 = w[6], r3 = w[7];
    const float sc24 = __uint_as_float(w[3]);
    const float ax = sc24 * ix, ay = sc24 * iy, az = sc24 * iz;
    const float bx = fmaf(__uint_as_float(w[0]), ix, nx), by = fmaf(__uint_as_float(w[1]), iy, ny), bz = fmaf(__uint_as_float(w[2]), iz, nz);
#define IRIS_SLABS80(D, C)                                                                                                        \
    {                                                                                                                             \
        const iris_h2 hx = __builtin_bit_cast(iris_h2, (uint32_t)w[8 + (C)]), hy = __builtin_bit_cast(iris_h2, (uint32_t)w[12 + (C)]), hz = __builtin_bit_cast(iris_h2, (uint32_t)wz[C]);   \
        float tn = fmaxf(fmaxf(fmaf((float)hx.x, ax, bx), fmaf((float)hy.x, ay, by)), fmaxf(fmaf((float)hz.x, az, bz), 0.f));      \
        float tf = fminf(fminf(fmaf((float)hx.y, ax, bx), fmaf((float)hy.y, ay, by)), fminf(fmaf((float)hz.y, az, bz), r.h.t));    \
        D = tf - tn;                                                                                                              \
    }
    float d0, d1, d2, d3;
    IRIS_SLABS80(d0, 0) IRIS_SLABS80(d1, 1) IRIS_SLABS80(d2, 2) IRIS_SLABS80(d3, 3)
#undef IRIS_SLABS80
#else
    iris_u16v w;
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(base), "s"(off_s) : "memory");
    uint32_t r0 = w[12], r1 = w[13], r2 = w[14], r3 = w[15];
    const float ax = __uint_as_float(w[3]) * ix, ay = __uint_as_float(w[4]) * iy, az = __uint_as_float(w[5]) * iz;
    const float bx = fmaf(__uint_as_float(w[0]), ix, nx), by = fmaf(__uint_as_float(w[1]), iy, ny), bz = fmaf(__uint_as_float(w[2]), iz, nz);
#if IRIS_SHARED_PAIRS
    // Bytes 0 / 2 and 1 / 3 of a plane word are isolated TOGETHER: w & 0x00ff00ff holds the planes of children 0 and 2 as the two f16 halves of one
    // scalar register, (w >> 8) & 0x00ff00ff those of children 1 and 3 -- on register PAIRS (the six plane words are adjacent), 9 scalar instructions
    // instead of the 24 single-byte extractions hipcc writes (asm: the compiler sees through the masks and goes back to single bytes).
    uint64_t m02[3], m13[3];
    const uint64_t kMask = 0x00ff00ff00ff00ffull;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        m13[k] = (uint64_t)w[6 + 2 * k] | ((uint64_t)w[7 + 2 * k] << 32);           // (in place: the word pair is dead afterwards)
        asm("s_and_b64 %0, %1, %2\n\ts_lshr_b64 %1, %1, 8\n\ts_and_b64 %1, %1, %2" : "=&s"(m02[k]), "+s"(m13[k]) : "s"(kMask) : "scc");
    }
    // words in table order: near x, near y | near z, far x | far y, far z
#define IRIS_H2(M, K, HI) __builtin_bit_cast(iris_h2, (uint32_t)((M)[K] >> ((HI) ? 32 : 0)))
#define IRIS_SLABS2(D, M, SEL)                                                                                                   \
    {                                                                                                                             \
        const iris_h2 qnx = IRIS_H2(M, 0, 0), qny = IRIS_H2(M, 0, 1), qnz = IRIS_H2(M, 1, 0), qfx = IRIS_H2(M, 1, 1), qfy = IRIS_H2(M, 2, 0), qfz = IRIS_H2(M, 2, 1); \
        float tn = fmaxf(fmaxf(fmaf((float)qnx.SEL, ax, bx), fmaf((float)qny.SEL, ay, by)), fmaxf(fmaf((float)qnz.SEL, az, bz), 0.f));   \
        float tf = fminf(fminf(fmaf((float)qfx.SEL, ax, bx), fmaf((float)qfy.SEL, ay, by)), fminf(fmaf((float)qfz.SEL, az, bz), r.h.t)); \
        D = tf - tn;                                                                                                              \
    }
    float d0, d1, d2, d3;
    IRIS_SLABS2(d0, m02, x) IRIS_SLABS2(d1, m13, x) IRIS_SLABS2(d2, m02, y) IRIS_SLABS2(d3, m13, y)
#undef IRIS_SLABS2
#undef IRIS_H2
#else
    const uint32_t nxq = w[6], nyq = w[7], nzq = w[8], fxq = w[9], fyq = w[10], fzq = w[11];
#define IRIS_PAIR(NQ, FQ, C) __builtin_bit_cast(iris_h2, (uint32_t)((((NQ) >> (8 * (C))) & 0xffu) | ((((FQ) >> (8 * (C))) & 0xffu) << 16)))
#define IRIS_SLABS(D, C)                                                                                                          \
    {                                                                                                                             \
        const iris_h2 hx = IRIS_PAIR(nxq, fxq, C), hy = IRIS_PAIR(nyq, fyq, C), hz = IRIS_PAIR(nzq, fzq, C);                       \
        float tn = fmaxf(fmaxf(fmaf((float)hx.x, ax, bx), fmaf((float)hy.x, ay, by)), fmaxf(fmaf((float)hz.x, az, bz), 0.f));      \
        float tf = fminf(fminf(fmaf((float)hx.y, ax, bx), fmaf((float)hy.y, ay, by)), fminf(fmaf((float)hz.y, az, bz), r.h.t));    \
        D = tf - tn;                                                                                                              \
    }
    float d0, d1, d2, d3;
    IRIS_SLABS(d0, 0) IRIS_SLABS(d1, 1) IRIS_SLABS(d2, 2) IRIS_SLABS(d3, 3)
#undef IRIS_SLABS
#undef IRIS_PAIR
#endif
#endif
    const int32_t b0 = __float_as_int(d0), b1 = __float_as_int(d1), b2 = __float_as_int(d2), b3 = __float_as_int(d3);
    const int32_t n01 = b0 & b1, n012 = n01 & b2;
    if (fast_push) {
        uint32_t c = b2 >= 0 ? r2 : r3;
        c = b1 >= 0 ? r1 : c;
        c = b0 >= 0 ? r0 : c;
        st.push3_fast(r3, b3 | n012, r2, b2 | n01, r1, b1 | b0);
        if ((n012 & b3) >= 0) r.cur = c;
        else r.cur = st.sp > 0 ? st.pop_lds() : kEmptyRef;
        return;
    }
    if ((n012 & b3) >= 0) {
        uint32_t c = b2 >= 0 ? r2 : r3;
        c = b1 >= 0 ? r1 : c;
        r.cur = b0 >= 0 ? r0 : c;
        if ((b3 | n012) >= 0) st.push(r3);
        if ((b2 | n01) >= 0) st.push(r2);
        if ((b1 | b0) >= 0) st.push(r1);
    } else {
        r.cur = st.sp > 0 ? st.pop() : kEmptyRef;
    }
}
#endif

// One triangle of the current leaf.
__device__ __forceinline__ RayXf leaf_phase_xform(const RayState& r) {
    RayXf x;
    x.ox = r.tox; x.oy = r.toy; x.oz = r.toz; x.sx = r.sx; x.sy = r.sy; x.sz = r.sz;
    x.offz = r.kz;                                   // 0 / 16 / 32
    x.offx = r.kz == 32u ? 0u : r.kz + 16u;
    x.offy = x.offx == 32u ? 0u : x.offx + 16u;
    return x;
}
template <class STACK>
__device__ __forceinline__ void leaf_step(const SceneDev& sc, RayState& r, const RayXf& xf, STACK& st, bool lds_only = false) {
    tri_test(sc, (int)((r.cur & 0x7fffffffu) >> 3), xf, r.h);
    r.cur += 7u;                                                  // leaf ref = leafbit | start << 3 | count: start + 1, count - 1
    if ((r.cur & 7u) == 0u) {
        if (lds_only) r.cur = st.sp > 0 ? st.pop_lds() : kEmptyRef;      // (wave-uniform: no lane of the step has entries beyond the LDS part)
        else r.cur = st.sp > 0 ? st.pop() : kEmptyRef;
    }
}

// -------------------------------------------------------------------------------------------------------
// LATENCY MODE of the one-ray-per-lane driver (round 5; template parameter JOINT of trace_bvh4: the host launches that instantiation of a kernel when the call does not fill the chip).
// A call of the path-tracing stages traces 262 144 rays (cfg 5): four waves per SIMD for ONE round, so the launch lasts as long as its longest wave, and that
// wave's length is its number of DEPENDENT memory round trips (tools/bench_pt_stage_scaling.py: 0.17-0.19 ms for any call of up to 131 072 rays).  The phase
// scheduling of trace_bvh4 serves issue throughput: an iteration is EITHER a node step or a triangle test, and the lanes in the other state wait -- a wave makes
// as many round trips as the two kinds of steps take one after the other.  Here every iteration first ISSUES the loads of both kinds (the four words of the node
// for the lanes at a node, the three planes of the triangle for the lanes at a leaf), waits for all of them together and then evaluates both: every lane with work
// advances in every iteration, one round trip each.  More instructions per ray (both streams are issued whenever any lane needs them), which a launch that leaves
// the vector ALUs idle does not pay for.  Same arithmetic per ray; the closest hit does not depend on the schedule (lexicographic minimum of (t, index)).
// -------------------------------------------------------------------------------------------------------
#if !IRIS_NODE80
template <class STACK>
__device__ __forceinline__ void node_eval_q8(RayState& r, STACK& st, const iris_u4v hd, const iris_u4v q1, const iris_u4v q2, const iris_u4v rf) {
    // (node_step's Q8 arithmetic on words that are already loaded; checked pushes)
    const float ix = r.ix, iy = r.iy, iz = r.iz, nx = r.nx, ny = r.ny, nz = r.nz;
    uint32_t r0 = rf.x, r1 = rf.y, r2 = rf.z, r3 = rf.w;
    const float ax = __uint_as_float(hd.w) * ix, ay = __uint_as_float(q1.x) * iy, az = __uint_as_float(q1.y) * iz;
    const float bx = fmaf(__uint_as_float(hd.x), ix, nx), by = fmaf(__uint_as_float(hd.y), iy, ny), bz = fmaf(__uint_as_float(hd.z), iz, nz);
    const uint32_t nxq = q1.z, nyq = q1.w, nzq = q2.x, fxq = q2.y, fyq = q2.z, fzq = q2.w;
    typedef _Float16 iris_h2 __attribute__((ext_vector_type(2)));
#define IRIS_PL(NQ, FQ, C) __builtin_bit_cast(iris_h2, __builtin_amdgcn_perm(NQ, FQ, 0x0c000c04u | ((uint32_t)(C) << 16) | (uint32_t)(C)))
#define IRIS_SL(D, C)                                                                                                             \
    {                                                                                                                             \
        const iris_h2 hx = IRIS_PL(nxq, fxq, C), hy = IRIS_PL(nyq, fyq, C), hz = IRIS_PL(nzq, fzq, C);                             \
        float tn = fmaxf(fmaxf(fmaf((float)hx.x, ax, bx), fmaf((float)hy.x, ay, by)), fmaxf(fmaf((float)hz.x, az, bz), 0.f));      \
        float tf = fminf(fminf(fmaf((float)hx.y, ax, bx), fmaf((float)hy.y, ay, by)), fminf(fmaf((float)hz.y, az, bz), r.h.t));    \
        D = tf - tn;                                                                                                              \
    }
    float d0, d1, d2, d3;
    IRIS_SL(d0, 0) IRIS_SL(d1, 1) IRIS_SL(d2, 2) IRIS_SL(d3, 3)
#undef IRIS_SL
#undef IRIS_PL
    const int32_t b0 = __float_as_int(d0), b1 = __float_as_int(d1), b2 = __float_as_int(d2), b3 = __float_as_int(d3);
    const int32_t n01 = b0 & b1, n012 = n01 & b2;
    if ((n012 & b3) >= 0) {
        uint32_t c = b2 >= 0 ? r2 : r3;
        c = b1 >= 0 ? r1 : c;
        r.cur = b0 >= 0 ? r0 : c;
        if ((b3 | n012) >= 0) st.push(r3);
        if ((b2 | n01) >= 0) st.push(r2);
        if ((b1 | b0) >= 0) st.push(r1);
    } else {
        r.cur = st.sp > 0 ? st.pop() : kEmptyRef;
    }
}
template <int LDS_DEPTH, bool GLOBAL_OVF>
__device__ __forceinline__ Hit trace_q8_joint(const SceneDev& sc, f3 o, f3 d, uint32_t* lds_stack, uint32_t* ovf) {
    RayState r;
    ray_begin(sc, r, o, d);
    Stack<LDS_DEPTH, GLOBAL_OVF> st; st.lds = (lds_u32*)lds_stack; st.ovf = ovf; st.sp = 0; st.tid = threadIdx.x;
    const RayXf xf = leaf_phase_xform(r);
    iris_u4v hd = {0u, 0u, 0u, 0u}, q1 = hd, q2 = hd, rf = hd;       // (defined once: a lane that does not load in an iteration keeps stale words nobody evaluates)
    iris_f4v X = {0.f, 0.f, 0.f, 0.f}, Y = X, Z = X;
    for (;;) {
        const bool at_node = r.cur != kEmptyRef && !(r.cur & kLeafBit);
        const bool at_leaf = r.cur != kEmptyRef && (r.cur & kLeafBit);
        if (__ballot(at_node || at_leaf) == 0) break;
        const int slot = (int)((r.cur & 0x7fffffffu) >> 3);
        if (at_node) {
            glb_u4v* n = (glb_u4v*)(reinterpret_cast<const char*>(sc.nodes) + (size_t)(uint32_t)(node_offset(r.cur) + r.oct_base));
            hd = n[0]; q1 = n[1]; q2 = n[2]; rf = n[3];
        }
        if (at_leaf) tri_load(sc, slot, xf, X, Y, Z);
        asm volatile("" : "+v"(hd), "+v"(q1), "+v"(q2), "+v"(rf), "+v"(X), "+v"(Y), "+v"(Z));      // both groups of loads issued before either is used: ONE round trip
        if (at_node) node_eval_q8(r, st, hd, q1, q2, rf);
        if (at_leaf) {
            tri_eval(X, Y, Z, slot, xf, r.h);
            r.cur += 7u;
            if ((r.cur & 7u) == 0u) r.cur = st.sp > 0 ? st.pop() : kEmptyRef;
        }
    }
    return r.h;
}
#endif

// One ray per lane, run to completion (primary rays, the path-tracing stages, the pixel-per-wave bake kernel).
template <int LAYOUT, bool COUNT = false, int LDS_DEPTH = kStackLds, bool GLOBAL_OVF = false, bool JOINT = false>
__device__ __forceinline__ Hit trace_bvh4(const SceneDev& sc, f3 o, f3 d, uint32_t* lds_stack, TraceStats* ts = nullptr, uint32_t* ovf = nullptr) {
#if !IRIS_NODE80
    if (JOINT && !COUNT && LAYOUT == kLayoutQ8) return trace_q8_joint<LDS_DEPTH, GLOBAL_OVF>(sc, o, d, lds_stack, ovf);      // (its own instantiation: 104 VGPRs against 77-83)
#endif
    RayState r;
    ray_begin(sc, r, o, d);
    Stack<LDS_DEPTH, GLOBAL_OVF> st; st.lds = (lds_u32*)lds_stack; st.ovf = ovf; st.sp = 0; st.tid = threadIdx.x;
    int max_sp = 0;
    const int kPhaseMinRt = sc.phase_min;
    for (;;) {
        // ---------------- node phase
        for (;;) {
            const bool at_node = r.cur != kEmptyRef && !(r.cur & kLeafBit);
            const int n_node = __popcll(__ballot(at_node));
            if (n_node == 0) break;
            if (n_node < kPhaseMinRt && __popcll(__ballot(r.cur != kEmptyRef && (r.cur & kLeafBit))) >= kPhaseMinRt) break;
            if (at_node) {
                if (COUNT) { ts->nodes++; ts->count_top(r.cur); ts->count_shared(r.cur, r.oct_base); if (first_active_lane()) ts->node_iters++; }
                node_step<LAYOUT>(sc, r, st);
                if (COUNT) max_sp = max(max_sp, st.sp);
            }
        }
        // ---------------- leaf phase (one triangle per iteration)
        const RayXf xf = leaf_phase_xform(r);
        for (;;) {
            const bool at_leaf = r.cur != kEmptyRef && (r.cur & kLeafBit);
            const int n_leaf = __popcll(__ballot(at_leaf));
            if (n_leaf == 0) break;
            if (n_leaf < kPhaseMinRt && __popcll(__ballot(r.cur != kEmptyRef && !(r.cur & kLeafBit))) >= kPhaseMinRt) break;
            if (at_leaf) {
                if (COUNT) { ts->tris++; if (first_active_lane()) ts->leaf_iters++; }
                leaf_step(sc, r, xf, st);
            }
        }
        if (__ballot(r.cur != kEmptyRef) == 0) break;
    }
    if (COUNT) { ts->sp_gt8 += max_sp > 8; ts->sp_gt12 += max_sp > 12; ts->sp_gt16 += max_sp > 16; }
    return r.h;
}

// -------------------------------------------------------------------------------------------------------
// Persistent-lane traversal ("dynamic fetch"): a lane whose ray is finished does not wait for the slowest ray of its wave;
// once kRefillMin lanes are idle they retire their hits and fetch new rays together.  Same node / leaf steps and phase
// scheduling as trace_bvh4, so per-ray results are identical; only which rays share a wave changes.
//   fetch(o, d)  -> bool : called by the idle lanes together: claim a ray and load its raw data into o / d; false = no ray left
//   prepare(o, d)        : turn the raw data into origin / direction
//   retire(h)            : called by a lane whose ray is complete, before it fetches again / at the end
// Measured on the 7-lobe bake: threshold 16: -1 %, 32: +4 %, 48: +7 %, 56: +4 %, 64 (= no refill): -4 %.
// A fetched ray is activated at once.  (Rounds 1-3 issued the loads in one round and activated the ray in the next, behind one node / leaf step of the
// other lanes: the six registers the raw data waited in and the extra loop exits cost more than the hidden latency was worth -- +1.2 % without.)
// -------------------------------------------------------------------------------------------------------
#ifndef IRIS_REFILL_MIN
#define IRIS_REFILL_MIN 48
#endif
constexpr int kRefillMin = IRIS_REFILL_MIN;
// STRAGGLER PARKING (round 6, -DIRIS_PARK=1; tools/bvh_eval/wavesim `wpark+tile`).  When a wave refills, the <= 16 rays it still carries are deep in the tree and share
// nothing with the 48 rays that start at the root: the fresh rays never reach the 44 lanes the shared scalar visit needs, and the wave keeps two populations apart
// for the rest of their lives.  With parking the refilling wave writes those rays' traversal state -- hit so far, current reference, the LDS part of the stack and the
// ray's id: 5 x 16 B -- to a WAVE-PRIVATE pool in the workspace (no exchange between waves: nothing to synchronise) and starts 64 fresh rays together; once the pool
// holds a wave's worth it is taken instead of fresh rays (the stragglers run with each other), and whatever is left is taken when the list is exhausted.  A ray whose
// stack has entries beyond the LDS part stays where it is.  Per-ray results do not change (the state is restored bit for bit; origin and direction are re-fetched).
#ifndef IRIS_PARK
#define IRIS_PARK 0
#endif
#ifndef IRIS_PARK_TAKE
#define IRIS_PARK_TAKE 64
#endif
#ifndef IRIS_PARK_TAIL           // no parking once fewer than this many rays are left in the tile's list, and the pool is taken from IRIS_PARK_TAIL_TAKE records on:
#define IRIS_PARK_TAIL 64        // the pools are then (nearly) empty when the list ends, instead of being drained at the end of the tile at a few lanes per wave
#endif
#ifndef IRIS_PARK_TAIL_TAKE
#define IRIS_PARK_TAIL_TAKE 64
#endif
constexpr int kParkCap = 128;            // records per wave (<= 63 waiting + 16 per refill round; 5 x 16 B each)
constexpr int kParkWords4 = 5;
// (records are addressed as a scalar base + a 32-bit per-lane byte offset through address-space-1 pointers, like the tables: a per-lane 64-bit pointer is two
//  registers that live across the refill round -- and were spilled to scratch there)
typedef __attribute__((address_space(1))) iris_u4v glb_u4v_rw;
__device__ __forceinline__ iris_u4v park_ld(const iris_u4v* base, uint32_t byte_off) { return *(glb_u4v*)(reinterpret_cast<const char*>(base) + (size_t)byte_off); }
__device__ __forceinline__ void park_st(iris_u4v* base, uint32_t byte_off, iris_u4v v) { *(glb_u4v_rw*)(reinterpret_cast<char*>(base) + (size_t)byte_off) = v; }
struct NoPark { static constexpr bool enabled = false; };
template <class GetId, class SetId, class Refetch, class Left, class Claim>
struct ParkOps {
    static constexpr bool enabled = true;
    iris_u4v* rec;        // this wave's records
    GetId get_id;         // the id of the ray this lane carries
    SetId set_id;         // ... set (an unparked ray)
    Refetch refetch;      // (id, o, d): issue the loads of a ray's raw origin / direction again
    Left left;            // rays left in the tile's list (wave-uniform)
    Claim claim;          // the idle lanes claim the next rays of the list together: -> ray id, or -1
};
template <int LAYOUT, bool COUNT, int LDS_DEPTH, bool GLOBAL_OVF, class Fetch, class Prepare, class Retire, class Park = NoPark>
__device__ __forceinline__ void trace_stream(const SceneDev& sc, uint32_t* lds_stack, uint32_t tid, uint32_t* ovf, TraceStats* ts, Fetch fetch, Prepare prepare,
                                             Retire retire, Park park = Park()) {
    RayState r;
    r.o = mk3(0.f, 0.f, 0.f); r.d = mk3(0.f, 0.f, 1.f);
    ray_begin(sc, r, r.o, r.d);
    r.cur = kEmptyRef;
    Stack<LDS_DEPTH, GLOBAL_OVF> st; st.lds = (lds_u32*)lds_stack; st.ovf = ovf; st.sp = 0; st.tid = tid;
    int shared_tries = kSharedTries;   // wave-uniform: misses the shared-visit test may still have before it is skipped until the next refill
    bool live = false;             // this lane holds a ray (in flight, or finished and not yet retired)
    bool more = true;              // wave-uniform: the ray list is not exhausted
    int n_pool = 0;                // wave-uniform: parked rays of this wave
    int max_sp = 0;
    const int kPhaseMinRt = sc.phase_min;
    for (;;) {
        // ---------------- retire finished rays, fetch and start new ones
        const bool idle0 = r.cur == kEmptyRef;
        const int n_idle = __popcll(__ballot(idle0));
        if (more && (n_idle >= kRefillMin || n_idle == __popcll(__ballot(1)))) {
            bool got = false;
            if constexpr (Park::enabled) {
                bool idle = idle0;
                // ---- park what is still in flight (while the list has a wave's worth of fresh rays left), then fill ALL lanes from one source
                const int left = park.left();
                if (left >= IRIS_PARK_TAIL) {
                    const bool can = !idle && st.sp <= LDS_DEPTH;
                    const unsigned long long mp = __ballot(can);
                    const int np = popc_mask(mp);
                    if (np > 0 && n_pool + np <= kParkCap) {
                        if (can) {
                            const int slot = n_pool + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mp, 0u));
                            const uint32_t qo = (uint32_t)slot * (kParkWords4 * 16u);
                            iris_u4v a0 = {__float_as_uint(r.h.t), __float_as_uint(r.h.u), __float_as_uint(r.h.v), (uint32_t)r.h.slot};
                            iris_u4v a1 = {(uint32_t)r.h.id, r.cur, (uint32_t)st.sp, (uint32_t)park.get_id()};
                            park_st(park.rec, qo, a0); park_st(park.rec, qo + 16u, a1);
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                iris_u4v e = {st.lds[(4 * k) * kBlock], st.lds[(4 * k + 1) * kBlock], st.lds[(4 * k + 2) * kBlock], st.lds[(4 * k + 3) * kBlock]};   // (entries beyond sp: garbage nobody reads)
                                asm volatile("" : "+v"(e));      // (four entries in flight at a time: the scheduler would otherwise read all twelve first)
                                park_st(park.rec, qo + 32u + 16u * k, e);
                            }
                            r.cur = kEmptyRef; live = false; idle = true;
                        }
                        n_pool += np;
                    }
                }
                const bool from_pool = n_pool >= IRIS_PARK_TAKE || (left < IRIS_PARK_TAIL && n_pool >= IRIS_PARK_TAIL_TAKE) || (left <= 0 && n_pool > 0);
                int uslot = -1;            // the record an unparked lane restores from (read in two steps: few registers live across ray_begin)
                int id = -1;               // the ray this lane starts: from the pool or from the list -- ONE place below issues its loads
                if (from_pool) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // the records were written by other LANES of this wave, rounds ago: their stores are waited for HERE, once per take, not at every parking
                    const unsigned long long mi = __ballot(idle);
                    if (idle) {
                        if (live) retire(r.h);
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mi, 0u));
                        if (rank < n_pool) {
                            uslot = n_pool - 1 - rank;
                            const uint32_t qo = (uint32_t)uslot * (kParkWords4 * 16u);
                            id = (int)park_ld(park.rec, qo + 16u).w;
#pragma unroll
                            for (int k = 0; k < 3; ++k) {      // the stack goes straight back to this lane's LDS column, four entries at a time
                                iris_u4v e = park_ld(park.rec, qo + 32u + 16u * k);
                                asm volatile("" : "+v"(e));
                                st.lds[(4 * k) * kBlock] = e.x; st.lds[(4 * k + 1) * kBlock] = e.y; st.lds[(4 * k + 2) * kBlock] = e.z; st.lds[(4 * k + 3) * kBlock] = e.w;
                            }
                        }
                    }
                    n_pool -= min(n_pool, popc_mask(mi));
                } else if (idle) {
                    if (live) retire(r.h);
                    id = park.claim();
                }
                f3 o_ = mk3(0.f, 0.f, 0.f), d_ = mk3(0.f, 0.f, 1.f);      // (locals, dead at the end of the round: as members of the loop-carried state they were spilled around the restore below)
                if (idle) {
                    live = got = id >= 0;
                    if (got) { park.set_id(id); park.refetch(id, o_, d_); }
                }
                if (__ballot(got) == 0 && n_pool == 0) more = false;
                shared_tries = kSharedTries;
                if (got) {
                    prepare(o_, d_);
                    ray_begin(sc, r, o_, d_);
                    st.sp = 0;
                    if (uslot >= 0) {
                        const uint32_t qo = (uint32_t)uslot * (kParkWords4 * 16u);
                        const iris_u4v a0 = park_ld(park.rec, qo), a1 = park_ld(park.rec, qo + 16u);
                        r.h.t = __uint_as_float(a0.x); r.h.u = __uint_as_float(a0.y); r.h.v = __uint_as_float(a0.z); r.h.slot = (int)a0.w;
                        r.h.id = (int)a1.x; r.cur = a1.y; st.sp = (int)a1.z;
                    }
                    if (COUNT) { ts->sp_gt8 += max_sp > 8; ts->sp_gt12 += max_sp > 12; ts->sp_gt16 += max_sp > 16; max_sp = 0; }
                }
            } else {
            if (idle0) {
                if (live) retire(r.h);
                live = got = fetch(r.o, r.d);
            }
            if (__ballot(got) == 0) more = false;
            shared_tries = kSharedTries;
            if (got) {
                prepare(r.o, r.d);
                ray_begin(sc, r, r.o, r.d);
                st.sp = 0;
                if (COUNT) { ts->sp_gt8 += max_sp > 8; ts->sp_gt12 += max_sp > 12; ts->sp_gt16 += max_sp > 16; max_sp = 0; }
            }
            }
        }
        if (__ballot(r.cur != kEmptyRef) == 0) {
            if (!more) break;
            continue;
        }
#if IRIS_EXP_NODRAIN
        // (UPPER-BOUND EXPERIMENT, wrong results: what would a tile cost without its drain?  When the list is exhausted the rays still in flight are dropped -- retired as misses --
        //  instead of being traversed to the end at falling lane utilisation.  Never defined in a shipped build.)
        if (!more) { r.h.slot = -1; r.h.u = r.h.v = 0.f; break; }
#endif
        // ---------------- node phase
        for (;;) {
            const bool at_node = r.cur != kEmptyRef && !(r.cur & kLeafBit);
            const unsigned long long m_node = __ballot(at_node);
#if IRIS_LOOP_NEST
            // (nested tests with an empty asm in front of each inner one: hipcc otherwise folds them into ONE boolean -- s_cselect / s_and / s_or on mask pairs and a
            //  v_cmp on a 64-bit count -- where each test is a scalar compare and a branch)
            const int n_node = popc_mask(m_node);
            if (n_node == 0) break;
            if (n_node < kPhaseMinRt) {
                asm volatile("");
                if (popc_mask(__ballot(r.cur != kEmptyRef && (r.cur & kLeafBit))) >= kPhaseMinRt) break;
            }
            if (n_node <= 64 - kRefillMin) {
                asm volatile("");
                if (more) {
                    asm volatile("");
                    if (popc_mask(__ballot(r.cur == kEmptyRef)) >= kRefillMin) break;   // enough idle lanes: go refill
                }
            }
#else
            const int n_node = __popcll(m_node);
            if (n_node == 0) break;
            if (n_node < kPhaseMinRt && __popcll(__ballot(r.cur != kEmptyRef && (r.cur & kLeafBit))) >= kPhaseMinRt) break;
            if ((!IRIS_IDLE_GATE || n_node <= 64 - kRefillMin) && more && popc_mask(__ballot(r.cur == kEmptyRef)) >= kRefillMin) break;   // enough idle lanes: go refill
#endif
#if IRIS_SCALAR_TOP
            if (LAYOUT == kLayoutQ8 && shared_tries > 0) {
                const uint32_t off = node_offset(r.cur) + r.oct_base;           // (meaningless in the lanes that are not at a node: masked out below)
                const uint32_t off0 = (uint32_t)__builtin_amdgcn_readlane((int)off, __builtin_ctzll(m_node));   // (m_node != 0 here: ctz, not ffs - 1 with its zero case)
                if (popc_mask(__ballot(off == off0) & m_node) >= IRIS_SCALAR_TOP) {     // (one compare; the masks are combined and counted on the scalar ALU)
                    if (COUNT) {      // (instrumented builds follow the SAME wave-level schedule as the timed kernel since round 5: the shared steps are taken and counted)
                        if (at_node) ts->count_shared(r.cur, r.oct_base);
                        if (at_node && off == off0) { ts->nodes++; ts->count_top(r.cur); if (!more) ts->drain_nodes++; }
                        if (first_active_lane()) { ts->node_iters++; ts->shared_path_iters++; if (!more) ts->drain_node_iters++; }
                    }
                    if (at_node && off == off0) node_step_shared(sc, r, st, off0, IRIS_FAST_PUSH && LDS_DEPTH >= 3 && __ballot(st.sp > LDS_DEPTH - 3) == 0);
                    if (COUNT) max_sp = max(max_sp, st.sp);
                    shared_tries = kSharedTries;
                    continue;
                }
                --shared_tries;       // the wave has diverged: after kSharedTries misses in a row the test is skipped until the next refill brings rays that start together
            }
#endif
            if (at_node) {
                if (COUNT) { ts->nodes++; ts->count_top(r.cur); ts->count_shared(r.cur, r.oct_base); if (first_active_lane()) ts->node_iters++; if (!more) { ts->drain_nodes++; if (first_active_lane()) ts->drain_node_iters++; } }
                const bool fast_push = IRIS_FAST_PUSH && LDS_DEPTH >= 3 && __ballot(st.sp > LDS_DEPTH - 3) == 0;
                if (COUNT && !fast_push && first_active_lane()) ts->slow_push_iters++;
                node_step<LAYOUT>(sc, r, st, fast_push);
                if (COUNT) max_sp = max(max_sp, st.sp);
            }
        }
        // ---------------- leaf phase
        const RayXf xf = leaf_phase_xform(r);
        for (;;) {
            const bool at_leaf = r.cur != kEmptyRef && (r.cur & kLeafBit);
#if IRIS_LOOP_NEST
            const int n_leaf = popc_mask(__ballot(at_leaf));
            if (n_leaf == 0) break;
            if (n_leaf < kPhaseMinRt) {
                asm volatile("");
                if (popc_mask(__ballot(r.cur != kEmptyRef && !(r.cur & kLeafBit))) >= kPhaseMinRt) break;
            }
            if (n_leaf <= 64 - kRefillMin) {
                asm volatile("");
                if (more) {
                    asm volatile("");
                    if (popc_mask(__ballot(r.cur == kEmptyRef)) >= kRefillMin) break;
                }
            }
#else
            const int n_leaf = __popcll(__ballot(at_leaf));
            if (n_leaf == 0) break;
            if (n_leaf < kPhaseMinRt && __popcll(__ballot(r.cur != kEmptyRef && !(r.cur & kLeafBit))) >= kPhaseMinRt) break;
            if ((!IRIS_IDLE_GATE || n_leaf <= 64 - kRefillMin) && more && popc_mask(__ballot(r.cur == kEmptyRef)) >= kRefillMin) break;
#endif
            if (at_leaf) {
                if (COUNT) { ts->tris++; if (first_active_lane()) ts->leaf_iters++; }
                leaf_step(sc, r, xf, st, IRIS_FAST_POP && __ballot(st.sp > LDS_DEPTH) == 0);
            }
        }
    }
    if (live) retire(r.h);
    if (COUNT) { ts->sp_gt8 += max_sp > 8; ts->sp_gt12 += max_sp > 12; ts->sp_gt16 += max_sp > 16; }
}

// Mitsuba Mesh::compute_surface_interaction restated: p = fma(p0,b0,fma(p1,b1,p2*b2)), b0 = (1-b1)-b2;
// n = normalize(cross(p1-p0,p2-p0)).
__device__ __forceinline__ void hit_vertices(const SceneDev& sc, const Hit& h, f3& p0, f3& p1, f3& p2) {
    const float4* r = sc.tris + (int64_t)h.slot * 4;
    float4 X = r[0], Y = r[1], Z = r[2];      // component-major: (p0.x,p1.x,p2.x,id) (p0.y,...) (p0.z,...)
    p0 = mk3(X.x, Y.x, Z.x); p1 = mk3(X.y, Y.y, Z.y); p2 = mk3(X.z, Y.z, Z.z);
}
__device__ __forceinline__ f3 hit_position(const Hit& h, f3 p0, f3 p1, f3 p2) {
    float b1 = h.u, b2 = h.v, b0 = (1.f - b1) - b2;
    return mk3(fmaf(p0.x, b0, fmaf(p1.x, b1, p2.x * b2)), fmaf(p0.y, b0, fmaf(p1.y, b1, p2.y * b2)),
               fmaf(p0.z, b0, fmaf(p1.z, b1, p2.z * b2)));
}
__device__ __forceinline__ f3 hit_normal(f3 p0, f3 p1, f3 p2) {
    f3 n = x_cross(sub3(p1, p0), sub3(p2, p0));
    float len = sqrtf(x_dot(n, n));
    return mk3(n.x / len, n.y / len, n.z / len);
}

}  // namespace iris
