// Microbenchmarks that calibrate the roofs the bake kernels are priced against (bench.py `roofline`).
// gfx950 only.  Prints one JSON object per measurement on stdout.
//
//   valu   : wave64 issue cost (shader cycles per wave-instruction per SIMD) of the instruction classes of node_step / tri_test,
//            at 1, 2, 4, 7, 8 waves per SIMD -> the VALU-issue roof
//   gather : lane-divergent global_load_dwordx4 of 64-B records (the BVH node / leaf-triangle access pattern): cycles per
//            wave-instruction per CU as a function of table size (L1 / L2 / Infinity Cache resident), loads per record (1, 3, 4 on the
//            same 64-B line), lanes sharing a record (coherence of sorted rays) and active lanes -> the vector-memory (TA / L1) roof
//   lds    : lane-divergent ds_read_b128 of 64-B records staged in LDS (AoS and chunk-planar layouts) -> what LDS-staged node
//            packets cost per visit
//
// Build: make -C tools/microbench      Run on the GPU box: tools/microbench/mb [valu|gather|lds|all]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static __device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
static __device__ __forceinline__ unsigned long long stamp_real() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// ------------------------------------------------------------------------------------------------ VALU issue
// 32 independent instructions per loop trip on 16 destination registers (two passes), sources that never change.
enum { OP_FMA = 0, OP_FMA_MIX, OP_PERM, OP_CNDMASK, OP_MAX, OP_MIN3, OP_CMP, OP_MUL, OP_ADD_U32, OP_LSHL_ADD, OP_RCP, OP_MED3, OP_PK_FMA, OP_CND_SGPR, OP_CND_DST, OP_CMP_CND, OP_CMP_E64, OP_BFI, OP_MINMAX, OP_AND_OR, OP_MOV, OP_CVT_UB0, OP_CVT_UB2, OP_CVT_U32, OP_BFE, OP_AND, OP_OR, OP_LSHL, OP_LSHR, OP_SUB_F32, OP_ADD_F32, OP_MIN_U32, OP_MAX3, OP_LDEXP, OP_MAD_U24, OP_MUL_U24, OP_CVT_F16, OP_CND_E64_VCC, OP_CMP_NOP_CND, OP_MAD_MIX_LO, OP_MUL_LO_U32, OP_MUL_HI_U32, OP_MAD_U64, OP_PK_FMA_F16, OP_PK_MAX_F16, OP_PK_ADD_F16, OP_CVT_PKRTZ, OP_PK_MIN_F16_PK_FMA, OP_N };
static const char* kOpName[OP_N] = {"v_fma_f32", "v_fma_mix_f32", "v_perm_b32", "v_cndmask_b32", "v_max_f32", "v_min3_f32", "v_cmp_lt_f32",
                                    "v_mul_f32", "v_add_u32", "v_lshl_add_u32", "v_rcp_f32", "v_med3_f32", "v_pk_fma_f32",
                                    "v_cndmask_b32_e64(sgpr mask)", "v_cndmask_b32(dst!=src)", "v_cmp+3xv_cndmask", "v_cmp_lt_f32_e64(sgpr dst)", "v_bfi_b32",
                                    "v_min_f32+v_max_f32", "v_and_or_b32", "v_mov_b32", "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte2", "v_cvt_f32_u32", "v_bfe_u32", "v_and_b32", "v_or_b32",
                                    "v_lshlrev_b32", "v_lshrrev_b32", "v_sub_f32", "v_add_f32", "v_min_u32", "v_max3_f32", "v_ldexp_f32", "v_mad_u32_u24", "v_mul_u32_u24",
                                    "v_cvt_f32_f16", "v_cndmask_b32_e64(vcc operand)", "v_cmp_e32+4 fma+v_cndmask_e32", "v_fma_mix_f32(f32 srcs)",
                                    "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32",
                                    "v_pk_fma_f16", "v_pk_max_f16", "v_pk_add_f16", "v_cvt_pkrtz_f16_f32", "v_pk_min_f16+v_pk_fma_f16"};

#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters, unsigned long long* cyc, unsigned long long* real) {
    float a[16];
    float2 p[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = (float)(threadIdx.x + k) * 1e-3f; p[k] = make_float2(a[k], a[k] + 1.f); }
    float b = 1.0001f + (float)threadIdx.x * 1e-9f, c = 0.5f;
    float2 pb = make_float2(b, b), pc = make_float2(c, c);
    uint32_t sel = 0x0c000c04u;
    unsigned long long smask = __ballot(threadIdx.x & 1), sm[4] = {0, 0, 0, 0};
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(b) : "vcc");      // VCC defined before the loop
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = stamp(), r0 = stamp_real();
    for (int i = 0; i < iters; ++i) {
#define EMIT(OP, k)                                                                                                              \
    if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                                     \
    else if (OP == OP_FMA_MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[k]) : "v"(b), "v"(c));      \
    else if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(sel));                            \
    else if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b));                               \
    else if (OP == OP_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                            \
    else if (OP == OP_MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                              \
    else if (OP == OP_CMP) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");                                \
    else if (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                            \
    else if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                        \
    else if (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(b));                               \
    else if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));                                                         \
    else if (OP == OP_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                              \
    else if (OP == OP_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));                          \
    else if (OP == OP_CND_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(smask));                  \
    else if (OP == OP_CND_DST) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[k]) : "v"(b), "v"(c));                          \
    else if (OP == OP_CMP_CND) { if ((k & 3) == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");           \
                                 else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b)); }                         \
    else if (OP == OP_CMP_E64) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sm[k & 3]) : "v"(a[k]), "v"(b));                    \
    else if (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(b), "v"(c));                                   \
    else if (OP == OP_MINMAX) { if (k & 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                               \
                                else asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b)); }                                   \
    else if (OP == OP_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                             \
    else if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(b));                                                  \
    else if (OP == OP_CVT_UB0) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[k]) : "v"(b));                                       \
    else if (OP == OP_CVT_UB2) asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(a[k]) : "v"(b));                                       \
    else if (OP == OP_CVT_U32) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[k]) : "v"(b));                                          \
    else if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[k]));                                                     \
    else if (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                              \
    else if (OP == OP_OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                                \
    else if (OP == OP_LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[k]));                                                   \
    else if (OP == OP_LSHR) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[k]));                                                   \
    else if (OP == OP_SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                          \
    else if (OP == OP_ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                          \
    else if (OP == OP_MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                          \
    else if (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                                \
    else if (OP == OP_LDEXP) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[k]) : "v"(sel));                                        \
    else if (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                          \
    else if (OP == OP_MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                      \
    else if (OP == OP_CVT_F16) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[k]) : "v"(b));                                          \
    else if (OP == OP_CND_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b));                         \
    else if (OP == OP_CMP_NOP_CND) { if ((k & 7) == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");      \
                                     else if ((k & 7) == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b));   \
                                     else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)); }                 \
    else if (OP == OP_MAD_MIX_LO) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                       \
    else if (OP == OP_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                    \
    else if (OP == OP_MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                    \
    else if (OP == OP_MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[k]) : "v"(b), "v"(c) : "vcc");               \
    else if (OP == OP_PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                          \
    else if (OP == OP_PK_MAX_F16) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                      \
    else if (OP == OP_PK_ADD_F16) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                      \
    else if (OP == OP_CVT_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(a[k]) : "v"(b), "v"(c));                        \
    else if (OP == OP_PK_MIN_F16_PK_FMA) { if (k & 1) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[k]) : "v"(b));                  \
                                           else asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)); }
#define ONE(k) EMIT(OP, k)
        R16(ONE) R16(ONE)
#undef ONE
    }
    const unsigned long long t1 = stamp(), r1 = stamp_real();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k] + p[k].x + p[k].y;
    if (s == 123.456f || (sm[0] ^ sm[1] ^ sm[2] ^ sm[3]) == 0x123456789ull) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        cyc[w] = t1 - t0; real[w] = r1 - r0;
    }
}

// Which instruction classes ISSUE TOGETHER (round 5): the same loop with two opcodes alternating, A on the even destination registers, B on the odd ones (all 32
// instructions of a trip independent of each other).  If A and B can share an issue slot the pair costs what the slower one costs alone; if not, the sum.
template <int OPA, int OPB>
__global__ __launch_bounds__(256) void pair_kernel(float* out, int iters, unsigned long long* cyc, unsigned long long* real) {
    float a[16];
    float2 p[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = (float)(threadIdx.x + k) * 1e-3f; p[k] = make_float2(a[k], a[k] + 1.f); }
    float b = 1.0001f + (float)threadIdx.x * 1e-9f, c = 0.5f;
    float2 pb = make_float2(b, b), pc = make_float2(c, c);
    uint32_t sel = 0x0c000c04u;
    unsigned long long smask = __ballot(threadIdx.x & 1), sm[4] = {0, 0, 0, 0};
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(b) : "vcc");
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = stamp(), r0 = stamp_real();
    for (int i = 0; i < iters; ++i) {
#define TWO(k) if ((k) & 1) { EMIT(OPB, k) } else { EMIT(OPA, k) }
        R16(TWO) R16(TWO)
#undef TWO
    }
    const unsigned long long t1 = stamp(), r1 = stamp_real();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k] + p[k].x + p[k].y;
    if (s == 123.456f || (sm[0] ^ sm[1] ^ sm[2] ^ sm[3]) == 0x123456789ull) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        cyc[w] = t1 - t0; real[w] = r1 - r0;
    }
    (void)pb; (void)pc; (void)sel; (void)smask;
}
template <int OPA, int OPB>
static void run_pair(int cus, int waves_per_simd, float* d_out, unsigned long long* d_cyc, unsigned long long* d_real) {
    const int iters = 4096, blocks = cus * waves_per_simd, n_waves = blocks * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((pair_kernel<OPA, OPB>), dim3(blocks), dim3(256), 0, 0, d_out, 64, d_cyc, d_real);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((pair_kernel<OPA, OPB>), dim3(blocks), dim3(256), 0, 0, d_out, iters, d_cyc, d_real);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(n_waves);
    CHECK(hipMemcpy(cyc.data(), d_cyc, n_waves * 8, hipMemcpyDeviceToHost));
    double c = 0;
    for (int i = 0; i < n_waves; ++i) c += (double)cyc[i];
    c /= n_waves;
    const double insts = (double)iters * 32.0;
    printf("{\"bench\": \"pair\", \"a\": \"%s\", \"b\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_PAIR_per_simd\": %.3f, \"chip_Gwinst_per_s\": %.1f}\n",
           kOpName[OPA], kOpName[OPB], waves_per_simd, 2.0 * c / (insts * waves_per_simd), insts * n_waves / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

template <int OP>
static void run_valu(int cus, int waves_per_simd, float* d_out, unsigned long long* d_cyc, unsigned long long* d_real) {
    const int iters = 4096, blocks = cus * waves_per_simd, n_waves = blocks * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 64, d_cyc, d_real);   // warm
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_cyc, d_real);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(n_waves), real(n_waves);
    CHECK(hipMemcpy(cyc.data(), d_cyc, n_waves * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(real.data(), d_real, n_waves * 8, hipMemcpyDeviceToHost));
    double c = 0, r = 0;
    for (int i = 0; i < n_waves; ++i) { c += (double)cyc[i]; r += (double)real[i]; }
    c /= n_waves; r /= n_waves;
    const double insts = (double)iters * 32.0;
    const double ghz = c / (r * 10.0);   // s_memrealtime ticks at 100 MHz
    printf("{\"bench\": \"valu\", \"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_inst_per_simd\": %.3f, \"clock_ghz\": %.3f, "
           "\"chip_Gwinst_per_s\": %.1f, \"kernel_ms\": %.3f}\n",
           kOpName[OP], waves_per_simd, c / (insts * waves_per_simd), ghz, insts * n_waves / (ms * 1e-3) / 1e9, ms);
    fflush(stdout);
}

// ------------------------------------------------------------------------------------------------ divergent 64-B record gathers
// Every lane draws a record index from an LCG keyed by (wave, lane / GROUP, trip): lanes of one group read the same record.
// LOADS = 16-B loads per record (consecutive chunks of the same 64-B line, as node_step / tri_test issue them).
// DEP: the next index also depends on the loaded data (a dependent chain as in traversal: one record in flight per lane).
template <int LOADS, bool DEP>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ table, uint32_t mask, int group_shift, unsigned long long active,
                                                      int iters, uint32_t* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t s = (wave * 64u + (uint32_t)(lane >> group_shift)) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const bool on = (active >> lane) & 1ull;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = stamp();
    if (on) {
        for (int i = 0; i < iters; ++i) {
            constexpr int U = DEP ? 1 : 4;     // independent records in flight per lane and trip
            uint4 v[U][LOADS];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s = s * 1664525u + 1013904223u;
                const uint32_t idx = (s >> 8) & mask;
                const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(table) + ((size_t)idx << 6));
#pragma unroll
                for (int k = 0; k < LOADS; ++k) v[u][k] = p[k];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < LOADS; ++k) acc ^= (v[u][k].x ^ v[u][k].y) ^ (v[u][k].z ^ v[u][k].w);
            if (DEP) s ^= acc & 0xffu;
        }
    }
    const unsigned long long t1 = stamp();
    if (acc == 0x12345678u) out[0] = acc;
    if (lane == 0) cyc[wave] = t1 - t0;
}

struct GatherCfg { const char* level; size_t bytes; int loads; int group_shift; unsigned long long active; bool dep; int waves_per_simd; };

template <int LOADS, bool DEP>
static double launch_gather(const GatherCfg& g, int cus, const uint4* d_table, uint32_t* d_out, unsigned long long* d_cyc, int iters, float* ms_out) {
    const int blocks = cus * g.waves_per_simd, n_waves = blocks * 4;
    const uint32_t mask = (uint32_t)(g.bytes / 64 - 1);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather_kernel<LOADS, DEP>), dim3(blocks), dim3(256), 0, 0, d_table, mask, g.group_shift, g.active, iters / 4, d_out, d_cyc);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_kernel<LOADS, DEP>), dim3(blocks), dim3(256), 0, 0, d_table, mask, g.group_shift, g.active, iters, d_out, d_cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(ms_out, e0, e1));
    std::vector<unsigned long long> cyc(n_waves);
    CHECK(hipMemcpy(cyc.data(), d_cyc, n_waves * 8, hipMemcpyDeviceToHost));
    double c = 0;
    for (int i = 0; i < n_waves; ++i) c += (double)cyc[i];
    return c / n_waves;
}

static void run_gather(const GatherCfg& g, int cus, const uint4* d_table, uint32_t* d_out, unsigned long long* d_cyc) {
    const int iters = g.dep ? 2048 : 1024;
    float ms = 0.f;
    double c = 0;
#define GO(L, D) c = launch_gather<L, D>(g, cus, d_table, d_out, d_cyc, iters, &ms)
    if (g.dep) { if (g.loads == 1) GO(1, true); else if (g.loads == 3) GO(3, true); else GO(4, true); }
    else       { if (g.loads == 1) GO(1, false); else if (g.loads == 3) GO(3, false); else GO(4, false); }
#undef GO
    const int U = g.dep ? 1 : 4;
    const double winst_per_wave = (double)iters * U * g.loads;
    const int waves_per_cu = g.waves_per_simd * 4;
    const int n_active = __builtin_popcountll(g.active);
    const double cyc_per_winst_cu = c / (winst_per_wave * waves_per_cu);       // cycles of the CU's vector-memory path per wave-instruction
    const double lane_loads = winst_per_wave * n_active * waves_per_cu * cus;  // 16-B lane loads, whole chip
    printf("{\"bench\": \"gather\", \"level\": \"%s\", \"table_bytes\": %zu, \"loads_per_record\": %d, \"lanes_per_record\": %d, \"active_lanes\": %d, "
           "\"dependent\": %s, \"waves_per_simd\": %d, \"cycles_per_wave_inst_per_cu\": %.2f, \"lane_loads_per_cycle_per_cu\": %.3f, "
           "\"chip_TBps_16B\": %.2f, \"cycles_per_record_visit_per_wave\": %.1f, \"kernel_ms\": %.3f}\n",
           g.level, g.bytes, g.loads, 1 << g.group_shift, n_active, g.dep ? "true" : "false", g.waves_per_simd, cyc_per_winst_cu,
           n_active / cyc_per_winst_cu, lane_loads * 16.0 / (ms * 1e-3) / 1e12, c / ((double)iters * U), ms);
    fflush(stdout);
}

// ------------------------------------------------------------------------------------------------ LDS-staged 64-B records
// LAYOUT 0: AoS (record r at r*64, chunks consecutive)   1: AoS padded to 80 B   2: chunk-planar (chunk k of record r at (k*N + r)*16)
template <int LAYOUT>
__global__ __launch_bounds__(256) void lds_kernel(int n_rec, int group_shift, int iters, uint32_t* out, unsigned long long* cyc) {
    extern __shared__ uint4 s_tab[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int words = LAYOUT == 1 ? n_rec * 5 : n_rec * 4;
    for (int i = threadIdx.x; i < words; i += 256) s_tab[i] = make_uint4(i, i * 3, i * 5, i * 7);
    __syncthreads();
    uint32_t s = (wave * 64u + (uint32_t)(lane >> group_shift)) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const unsigned long long t0 = stamp();
    for (int i = 0; i < iters; ++i) {
        uint4 v[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            s = s * 1664525u + 1013904223u;
            const uint32_t idx = ((s >> 8) & 0xffffu) % (uint32_t)n_rec;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[u][k] = LAYOUT == 0 ? s_tab[idx * 4 + k] : LAYOUT == 1 ? s_tab[idx * 5 + k] : s_tab[k * n_rec + idx];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc ^= (v[u][k].x ^ v[u][k].y) ^ (v[u][k].z ^ v[u][k].w);
    }
    const unsigned long long t1 = stamp();
    if (acc == 0x12345678u) out[0] = acc;
    if (lane == 0) cyc[wave] = t1 - t0;
}

template <int LAYOUT>
static void run_lds(int cus, int n_rec, int group_shift, int blocks_per_cu, uint32_t* d_out, unsigned long long* d_cyc) {
    const int iters = 2048, blocks = cus * blocks_per_cu, n_waves = blocks * 4;
    const size_t lds = (size_t)n_rec * (LAYOUT == 1 ? 80 : 64);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_kernel<LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lds_kernel<LAYOUT>, dim3(blocks), dim3(256), lds, 0, n_rec, group_shift, 64, d_out, d_cyc);
    hipLaunchKernelGGL(lds_kernel<LAYOUT>, dim3(blocks), dim3(256), lds, 0, n_rec, group_shift, iters, d_out, d_cyc);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> cyc(n_waves);
    CHECK(hipMemcpy(cyc.data(), d_cyc, n_waves * 8, hipMemcpyDeviceToHost));
    double c = 0;
    for (int i = 0; i < n_waves; ++i) c += (double)cyc[i];
    c /= n_waves;
    const double winst = (double)iters * 2 * 4;
    printf("{\"bench\": \"lds\", \"layout\": \"%s\", \"records\": %d, \"lanes_per_record\": %d, \"waves_per_cu\": %d, "
           "\"cycles_per_ds_read_b128_per_cu\": %.2f, \"cycles_per_record_visit_per_cu\": %.1f}\n",
           LAYOUT == 0 ? "aos64" : LAYOUT == 1 ? "aos80" : "planar", n_rec, 1 << group_shift, blocks_per_cu * 4, c / (winst * blocks_per_cu * 4),
           c / ((double)iters * 2 * blocks_per_cu * 4));
    fflush(stdout);
}

// ------------------------------------------------------------------------------------------------ counter calibration (round 3)
// One process = one known workload, so that a `rocprofv3 --pmc ... -- tools/microbench/mb calib <what>` run can be compared with ground truth:
//   calib fma | mix | blend | max | rcp : the valu kernel of that class, 7 waves/SIMD, 4096 x 32 instructions per wave -> what SQ_INSTS_VALU,
//                                   SQ_ACTIVE_INST_VALU and SQ_ACTIVE_INST_VALU2 read for a pure dual-issue class, a pure 4-cycle class and the mix
//   calib gather | gather1        : lane-divergent random 64-B records of a 1 GiB table (4 x 16 B / 1 x 16 B per record): known bytes requested
//   calib stream                  : every lane reads consecutive 16 B (wave = 1 KiB contiguous), 1 GiB once: the guide's 1/2-counting case
// Prints the ground truth (instructions / bytes of the timed dispatch) as JSON; tools/calib_summary.py joins it with the counter CSVs.
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ table, size_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = table[i];
        acc ^= (v.x ^ v.y) ^ (v.z ^ v.w);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int OP>
static void calib_valu(int cus, float* d_out, unsigned long long* d_cyc, unsigned long long* d_real) {
    const int w = 7, iters = 4096, blocks = cus * w, n_waves = blocks * 4;
    run_valu<OP>(cus, w, d_out, d_cyc, d_real);     // dispatch 1 = 64-trip warm-up, dispatch 2 = the measured one
    printf("{\"bench\": \"calib\", \"what\": \"valu\", \"op\": \"%s\", \"waves\": %d, \"loop_wave_insts\": %.0f, \"dispatch\": 2}\n", kOpName[OP], n_waves,
           (double)iters * 32.0 * n_waves);
}

static void calib_mem(const char* what, int cus, uint32_t* d_out, unsigned long long* d_cyc) {
    const size_t bytes = 1ull << 30;
    uint4* d_table;
    CHECK(hipMalloc(&d_table, bytes));
    CHECK(hipMemset(d_table, 0x5a, bytes));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    if (!strcmp(what, "stream")) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_kernel, dim3(cus * 8), dim3(256), 0, 0, d_table, bytes / 16, d_out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"bench\": \"calib\", \"what\": \"stream\", \"bytes_requested\": %zu, \"kernel_ms\": %.3f, \"TBps\": %.3f, \"dispatch\": 1}\n", bytes, ms,
               bytes / (ms * 1e-3) / 1e12);
    } else {
        const int loads = !strcmp(what, "gather1") ? 1 : 4;
        GatherCfg g{"HBM1G", bytes, loads, 0, ~0ull, false, 7};
        const int iters = 1024, blocks = cus * g.waves_per_simd, n_waves = blocks * 4;
        if (loads == 1) launch_gather<1, false>(g, cus, d_table, d_out, d_cyc, iters, &ms);
        else launch_gather<4, false>(g, cus, d_table, d_out, d_cyc, iters, &ms);
        const double records = (double)n_waves * 64.0 * iters * 4.0;     // U = 4 independent records per trip
        printf("{\"bench\": \"calib\", \"what\": \"%s\", \"records\": %.0f, \"bytes_requested\": %.0f, \"bytes_of_lines_touched_64B\": %.0f, "
               "\"bytes_of_lines_touched_128B\": %.0f, \"kernel_ms\": %.3f, \"dispatch\": 2, \"note\": \"dispatch 1 is the same kernel with iters/4\"}\n",
               what, records, records * 16.0 * loads, records * 64.0, records * 128.0, ms);
    }
    CHECK(hipFree(d_table));
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    const bool all = !strcmp(what, "all");
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"bench\": \"device\", \"name\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.name, prop.gcnArchName, cus, prop.clockRate / 1000);
    float* d_out; unsigned long long *d_cyc, *d_real;
    CHECK(hipMalloc(&d_out, 4096));
    CHECK(hipMalloc(&d_cyc, 8 * cus * 8 * 4 * 2));
    CHECK(hipMalloc(&d_real, 8 * cus * 8 * 4 * 2));
    if (!strcmp(what, "calib")) {
        const char* sub = argc > 2 ? argv[2] : "fma";
        if (!strcmp(sub, "fma")) calib_valu<OP_FMA>(cus, d_out, d_cyc, d_real);
        else if (!strcmp(sub, "mix")) calib_valu<OP_FMA_MIX>(cus, d_out, d_cyc, d_real);
        else if (!strcmp(sub, "blend")) calib_valu<OP_CMP_NOP_CND>(cus, d_out, d_cyc, d_real);
        else if (!strcmp(sub, "max")) calib_valu<OP_MAX>(cus, d_out, d_cyc, d_real);
        else if (!strcmp(sub, "rcp")) calib_valu<OP_RCP>(cus, d_out, d_cyc, d_real);
        else calib_mem(sub, cus, (uint32_t*)d_out, d_cyc);
        return 0;
    }

    if (all || !strcmp(what, "valu")) {
        for (int w : {2, 7}) {
            run_valu<OP_FMA>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_FMA_MIX>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_PERM>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CNDMASK>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MAX>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MIN3>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CMP>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MUL>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_ADD_U32>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_LSHL_ADD>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_RCP>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MED3>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_PK_FMA>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CND_SGPR>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CND_DST>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CMP_CND>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CMP_E64>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_BFI>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MINMAX>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_AND_OR>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MOV>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CVT_UB0>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CVT_UB2>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CVT_U32>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_BFE>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_AND>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_OR>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_LSHL>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_LSHR>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_SUB_F32>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_ADD_F32>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MIN_U32>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MAX3>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_LDEXP>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MAD_U24>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MUL_U24>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CVT_F16>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CND_E64_VCC>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CMP_NOP_CND>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MAD_MIX_LO>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MUL_LO_U32>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MUL_HI_U32>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_MAD_U64>(cus, w, d_out, d_cyc, d_real);
        }
    }
    if (!strcmp(what, "pair")) {          // which instruction classes issue together (round 5): A alternating with B, all independent, 7 waves per SIMD
#define PAIRS_WITH(A) run_pair<A, OP_FMA>(cus, 7, d_out, d_cyc, d_real); run_pair<A, OP_FMA_MIX>(cus, 7, d_out, d_cyc, d_real); run_pair<A, OP_PERM>(cus, 7, d_out, d_cyc, d_real); \
                      run_pair<A, OP_MAX3>(cus, 7, d_out, d_cyc, d_real); run_pair<A, OP_MAX>(cus, 7, d_out, d_cyc, d_real); run_pair<A, OP_SUB_F32>(cus, 7, d_out, d_cyc, d_real);   \
                      run_pair<A, OP_AND>(cus, 7, d_out, d_cyc, d_real); run_pair<A, OP_ADD_U32>(cus, 7, d_out, d_cyc, d_real); run_pair<A, OP_LSHL_ADD>(cus, 7, d_out, d_cyc, d_real); \
                      run_pair<A, OP_MOV>(cus, 7, d_out, d_cyc, d_real); run_pair<A, OP_CND_SGPR>(cus, 7, d_out, d_cyc, d_real); run_pair<A, OP_CMP_E64>(cus, 7, d_out, d_cyc, d_real); \
                      run_pair<A, OP_MUL>(cus, 7, d_out, d_cyc, d_real);
        PAIRS_WITH(OP_FMA_MIX) PAIRS_WITH(OP_FMA) PAIRS_WITH(OP_PERM) PAIRS_WITH(OP_MAX3) PAIRS_WITH(OP_AND) PAIRS_WITH(OP_LSHL_ADD) PAIRS_WITH(OP_CND_SGPR)
#undef PAIRS_WITH
    }
    if (!strcmp(what, "pk16")) {          // packed half arithmetic (round 4: would slab tests in packed f16 issue faster than v_fma_mix_f32?)
        for (int w : {2, 7}) {
            run_valu<OP_FMA_MIX>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_PK_FMA_F16>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_PK_MAX_F16>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_PK_ADD_F16>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_CVT_PKRTZ>(cus, w, d_out, d_cyc, d_real);
            run_valu<OP_PK_MIN_F16_PK_FMA>(cus, w, d_out, d_cyc, d_real);
        }
    }
    if (all || !strcmp(what, "gather")) {
        const size_t max_bytes = 128ull << 20;
        uint4* d_table;
        CHECK(hipMalloc(&d_table, max_bytes));
        std::vector<uint32_t> h(max_bytes / 4);
        uint32_t x = 1;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
        CHECK(hipMemcpy(d_table, h.data(), max_bytes, hipMemcpyHostToDevice));
        const unsigned long long ALL = ~0ull;
        const unsigned long long HALF = 0x5555555555555555ull;                  // every other lane
        const unsigned long long Q38 = 0x00000000ffffff00ull | 0x3fffull << 40; // 38 lanes (the kernel's ~0.6 utilisation), contiguous runs
        std::vector<GatherCfg> cfgs;
        struct Lv { const char* n; size_t b; };
        for (Lv lv : {Lv{"L1", 16u << 10}, Lv{"L2", 1u << 20}, Lv{"MALL16", 16u << 20}, Lv{"MALL80", 128u << 20}}) {
            for (int loads : {1, 3, 4}) {
                cfgs.push_back({lv.n, lv.b, loads, 0, ALL, false, 7});
                cfgs.push_back({lv.n, lv.b, loads, 0, ALL, true, 7});
            }
            for (int gs : {1, 2, 4, 6}) cfgs.push_back({lv.n, lv.b, 4, gs, ALL, false, 7});
            cfgs.push_back({lv.n, lv.b, 4, 0, HALF, false, 7});
            cfgs.push_back({lv.n, lv.b, 4, 0, Q38, false, 7});
            cfgs.push_back({lv.n, lv.b, 4, 1, Q38, true, 7});
            cfgs.push_back({lv.n, lv.b, 4, 0, ALL, false, 2});
            cfgs.push_back({lv.n, lv.b, 4, 0, ALL, true, 2});
        }
        for (const auto& g : cfgs) run_gather(g, cus, d_table, (uint32_t*)d_out, d_cyc);
        CHECK(hipFree(d_table));
    }
    if (all || !strcmp(what, "lds")) {
        for (int n_rec : {85, 341}) {
            for (int gs : {0, 1, 2, 4}) {
                for (int bpc : {1, 4}) {
                    run_lds<0>(cus, n_rec, gs, bpc, (uint32_t*)d_out, d_cyc);
                    run_lds<1>(cus, n_rec, gs, bpc, (uint32_t*)d_out, d_cyc);
                    run_lds<2>(cus, n_rec, gs, bpc, (uint32_t*)d_out, d_cyc);
                }
            }
        }
    }
    return 0;
}
