// Static instruction count of a PER-LANE BVH8 node visit next to the BVH4 visit of the bake kernels (round 5, verdict item 1a) -- compiled, never run:
//   tools/bvh_eval/node_step8_isa.sh   (hipcc -S with the flags of iris_amd/csrc/Makefile, counts the vector instructions between the loop labels)
// The BVH8 step is written the way the BVH4 step of iris_trace.h is: a 128-B node per ray octant, children pre-ordered front to back, 8-bit planes read as f16
// subnormals by v_fma_mix_f32 (v_perm_b32 pairs near / far), hit tests from the sign bits of the slab-interval lengths, the first child hit is next, the others
// are written unconditionally at the top of the lane's LDS stack and the top moves by the hit bit (push3_fast's scheme with seven entries).
#include <hip/hip_runtime.h>
#include "../../iris_amd/csrc/iris_trace.h"

using namespace iris;

// {origin.xyz, scale.x} {scale.y, scale.z, -, -} {near_x[0..3], near_x[4..7], near_y[0..3], near_y[4..7]} {near_z lo, hi, far_x lo, hi} {far_y lo, hi, far_z lo, hi} {ref[0..3]} {ref[4..7]}
template <class STACK>
__device__ __forceinline__ void node_step8(const SceneDev& sc, RayState& r, STACK& st) {
    const float ix = r.ix, iy = r.iy, iz = r.iz, nx = r.nx, ny = r.ny, nz = r.nz;
    glb_u4v* n = (glb_u4v*)(reinterpret_cast<const char*>(sc.nodes) + (size_t)(uint32_t)((r.cur << 7) + r.oct_base));
    const iris_u4v hd = n[0], h2 = n[1], q1 = n[2], q2 = n[3], q3 = n[4], ra = n[5], rb = n[6];
    uint32_t rf[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
    const float ax = __uint_as_float(hd.w) * ix, ay = __uint_as_float(h2.x) * iy, az = __uint_as_float(h2.y) * iz;
    const float bx = fmaf(__uint_as_float(hd.x), ix, nx), by = fmaf(__uint_as_float(hd.y), iy, ny), bz = fmaf(__uint_as_float(hd.z), iz, nz);
    typedef _Float16 iris_h2 __attribute__((ext_vector_type(2)));
#define P8(NQ, FQ, C) __builtin_bit_cast(iris_h2, __builtin_amdgcn_perm(NQ, FQ, 0x0c000c04u | ((uint32_t)(C) << 16) | (uint32_t)(C)))
#define SLAB8(D, NX, FX, NY, FY, NZ, FZ, C)                                                                                       \
    {                                                                                                                             \
        const iris_h2 hx = P8(NX, FX, C), hy = P8(NY, FY, C), hz = P8(NZ, FZ, C);                                                  \
        float tn = fmaxf(fmaxf(fmaf((float)hx.x, ax, bx), fmaf((float)hy.x, ay, by)), fmaxf(fmaf((float)hz.x, az, bz), 0.f));      \
        float tf = fminf(fminf(fmaf((float)hx.y, ax, bx), fmaf((float)hy.y, ay, by)), fminf(fmaf((float)hz.y, az, bz), r.h.t));    \
        D = tf - tn;                                                                                                              \
    }
    float d[8];
    SLAB8(d[0], q1.x, q2.z, q1.z, q3.x, q2.x, q3.z, 0) SLAB8(d[1], q1.x, q2.z, q1.z, q3.x, q2.x, q3.z, 1)
    SLAB8(d[2], q1.x, q2.z, q1.z, q3.x, q2.x, q3.z, 2) SLAB8(d[3], q1.x, q2.z, q1.z, q3.x, q2.x, q3.z, 3)
    SLAB8(d[4], q1.y, q2.w, q1.w, q3.y, q2.y, q3.w, 0) SLAB8(d[5], q1.y, q2.w, q1.w, q3.y, q2.y, q3.w, 1)
    SLAB8(d[6], q1.y, q2.w, q1.w, q3.y, q2.y, q3.w, 2) SLAB8(d[7], q1.y, q2.w, q1.w, q3.y, q2.y, q3.w, 3)
#undef SLAB8
#undef P8
    int32_t b[8], none[8];                      // sign set: child j not hit; none[j]: none of children 0 .. j-1 hit
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = __float_as_int(d[j]);
    none[0] = (int32_t)0x80000000;
#pragma unroll
    for (int j = 1; j < 8; ++j) none[j] = none[j - 1] & b[j - 1];
    asm volatile("" : "+v"(rf[0]), "+v"(rf[1]), "+v"(rf[2]), "+v"(rf[3]), "+v"(rf[4]), "+v"(rf[5]), "+v"(rf[6]), "+v"(rf[7]));
    uint32_t c = rf[7];
#pragma unroll
    for (int j = 6; j >= 0; --j) c = b[j] >= 0 ? rf[j] : c;
    // pushes, farthest first: child j is pushed iff it is hit and a nearer child is hit too
    int u = st.sp;
#pragma unroll
    for (int j = 7; j >= 1; --j) { st.lds[u * kBlock] = rf[j]; u += 1 + ((b[j] | none[j]) >> 31); }
    st.sp = u;
    if ((none[7] & b[7]) >= 0) r.cur = c;
    else r.cur = st.sp > 0 ? st.pop_lds() : kEmptyRef;
}

// minimal drivers: the node loop of trace_bvh4 without the leaf phase, so that the loop body is the visit
template <int W>
__global__ __launch_bounds__(kBlock, 7) void visit_kernel(SceneDev sc, const float* rays, int n, uint32_t* out) {
    __shared__ uint32_t s_stack[12 * kBlock];
    RayState r;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    ray_begin(sc, r, ld3(rays + (int64_t)i * 6), ld3(rays + (int64_t)i * 6 + 3));
    Stack<12, true> st; st.lds = (lds_u32*)(s_stack + threadIdx.x); st.ovf = out; st.sp = 0; st.tid = threadIdx.x;
    for (int k = 0; k < n; ++k) {
        asm volatile("; VISIT_BEGIN");
        if (W == 4) node_step<kLayoutQ8>(sc, r, st, true);
        else node_step8(sc, r, st);
        asm volatile("; VISIT_END");
        if (r.cur == kEmptyRef) r.cur = (uint32_t)k;
    }
    out[i] = r.cur + (uint32_t)st.sp;
}
template __global__ void visit_kernel<4>(SceneDev, const float*, int, uint32_t*);
template __global__ void visit_kernel<8>(SceneDev, const float*, int, uint32_t*);
