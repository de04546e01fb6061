// The material network of the reference's refine / emitter-training stages, inference only: NGPBRDF.forward (model/brdf.py:213-260) =
// tiny-cuda-nn NetworkWithInputEncoding(3 -> HashGrid{32 levels x 2 features, 2^19 entries, base 16, x 1.3} -> FullyFusedMLP{64 x 2 hidden, ReLU} -> 5)
// + sigmoid.  tiny-cuda-nn is third party and CUDA only: what is implemented is its published algorithm (Mueller et al. 2022, section 3), restated
// for the tests in oracle/ngp_torch.py ("parity unpinned").
//
//   ngp_encode_kernel  one thread per (point, level), blockIdx.y = level: the blocks of a level are dispatched together, so the 8 x 4-B corner gathers
//                      of a wave go to ONE level's table (<= 2 MiB of half2 entries: it lives in an XCD's 4 MiB L2 while the level is being worked on).
//                      Features go to a [level][point] half2 plane: 256 contiguous bytes per wave-store.  Gather-bound (L2 / Infinity Cache lines).
//   ngp_mlp_kernel     the 64 -> 64 -> 64 -> 16 perceptron on the matrix cores: v_mfma_f32_32x32x16_f16, one wave per 32 points, the weights of all
//                      three layers resident in registers as A fragments, the activations handed from one layer's accumulators to the next layer's B
//                      operand WITHOUT leaving the registers (a 32x32 f32 accumulator tile has the point on the lane and the neuron in the register
//                      index; the next layer's weights are loaded in the matching permuted k order).  This IS a dense contraction (the bake path is not).
#pragma once
#include "iris_device.h"

namespace iris {

constexpr int kNgpLevels = 32, kNgpWidth = 64, kNgpOutPad = 16, kNgpOut = 5;
constexpr int kNgpMlpParams = kNgpWidth * 64 + kNgpWidth * kNgpWidth + kNgpOutPad * kNgpWidth;   // 9216 halves: W1 (64 x 64), W2 (64 x 64), W3 (16 x 64), row-major (out x in)

struct NgpLevels {
    float scale[kNgpLevels];
    uint32_t res[kNgpLevels];
    uint32_t size[kNgpLevels];     // table entries of the level
    uint32_t offset[kNgpLevels];   // first entry of the level in the table
};
struct NgpArgs {
    NgpLevels lv;
    const uint32_t* grid;          // half2 entries (two features), all levels
    const _Float16* w;             // kNgpMlpParams halves
    const float* pos;              // (N, 3) world space
    uint32_t* feat;                // [level][n_chunk] half2
    float* albedo; float* rough; float* metal;   // (N,3), (N), (N)
    int64_t n0;                    // first point of this chunk
    int n;                         // points in this chunk
    int n_chunk;                   // plane stride of feat
    float vmin, den;               // voxel_min, float32(voxel_max - voxel_min)
};

typedef _Float16 iris_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 iris_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 iris_h2v __attribute__((ext_vector_type(2)));
typedef float iris_f16v __attribute__((ext_vector_type(16)));
typedef uint32_t ngp_u2a __attribute__((ext_vector_type(2), aligned(4)));     // two adjacent table entries, 4-byte aligned

// tiny-cuda-nn grid.h, kernel_grid, restated: position -> cell + weights, 8 corners, dense index while it fits the table, coherent prime hash otherwise
__global__ __launch_bounds__(256) void ngp_encode_kernel(NgpArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int level = blockIdx.y;
    if (i >= a.n) return;
    const float scale = a.lv.scale[level];
    const uint32_t res = a.lv.res[level], size = a.lv.size[level];
    const uint32_t* table = a.grid + a.lv.offset[level];
    const float* pp = a.pos + (a.n0 + i) * 3;
    float w[3]; uint32_t cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        // model/brdf.py:252-254: (position - voxel_min) / (voxel_max - voxel_min), then * 2 - 1
        const float x = (pp[d] - a.vmin) / a.den * 2.0f - 1.0f;
        const float p = fmaf(scale, x, 0.5f);
        const float fl = floorf(p);
        w[d] = p - fl;
        cell[d] = (uint32_t)(int)fl;                  // negative cells wrap, as in the library
    }
    // dense indexing covers as many dimensions as fit the table (wave-uniform: a property of the level)
    uint32_t stride1 = 0, stride2 = 0; bool hashed;
    {
        uint64_t s = 1; int dims = 0;
        uint32_t st[3] = {0, 0, 0};
        for (int d = 0; d < 3 && s <= size; ++d) { st[d] = (uint32_t)s; s *= res; ++dims; }
        stride1 = st[1]; stride2 = st[2];
        hashed = (uint64_t)size < s;
        if (dims < 3 && !hashed) hashed = true;       // (cannot happen: the loop only stops early once the stride exceeds the table)
    }
    _Float16 acc0 = (_Float16)0.f, acc1 = (_Float16)0.f;
    // The 8 corners in the library's order (x fastest), two x-neighbours at a time.  What bounds this kernel is the number of 64-B requests its gathers send
    // to the L2s (profiles/r4_ngp_pmc.json), and the two x-neighbours of a cell are adjacent table entries whenever the indexing lets them be -- always on a dense
    // level, for even x on a hashed one (x ^ K and (x + 1) ^ K differ in bit 0 only) --: one 8-byte load then fetches both.
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
        uint32_t g1 = (c & 2) ? cell[1] + 1u : cell[1], g2 = (c & 4) ? cell[2] + 1u : cell[2];
        uint32_t ia, ib;
        if (hashed) { const uint32_t k = (g1 * 2654435761u) ^ (g2 * 805459861u); ia = (cell[0] * 1u) ^ k; ib = ((cell[0] + 1u) * 1u) ^ k; }
        else { const uint32_t k = g1 * stride1 + g2 * stride2; ia = cell[0] + k; ib = cell[0] + 1u + k; }
        ia %= size; ib %= size;
        uint32_t ra, rb;
        // (the pair's base is an ODD entry about half the time on dense levels: the 8-byte load goes through a vector type declared 4-byte aligned -- gfx950
        //  global memory runs in unaligned-access mode, so it is still ONE global_load_dwordx2, and no C++ alignment rule is broken)
        if (ib == ia + 1u) { const ngp_u2a v = *reinterpret_cast<const ngp_u2a*>(table + ia); ra = v.x; rb = v.y; }
        else if (ia == ib + 1u) { const ngp_u2a v = *reinterpret_cast<const ngp_u2a*>(table + ib); ra = v.y; rb = v.x; }
        else { ra = table[ia]; rb = table[ib]; }
#pragma unroll
        for (int xx = 0; xx < 2; ++xx) {
            float wgt = 1.0f;
            wgt = xx ? wgt * w[0] : wgt * (1.0f - w[0]);
            wgt = (c & 2) ? wgt * w[1] : wgt * (1.0f - w[1]);
            wgt = (c & 4) ? wgt * w[2] : wgt * (1.0f - w[2]);
            const iris_h2v v = __builtin_bit_cast(iris_h2v, xx ? rb : ra);
            // result += (half)(weight * value): every term rounded to half, the sum a half add.  The product is rounded to f32 FIRST and then to half, as a
            // C compiler for any other target does it: behind the barrier hipcc cannot fold the multiplication into v_fma_mixlo_f16, which rounds the exact
            // product to half once -- measured different from the two-step rounding in 1.4e-4 of the features (tests/test_ngp.py compares bit for bit).
            float t0 = wgt * (float)v.x, t1 = wgt * (float)v.y;
            asm volatile("" : "+v"(t0), "+v"(t1));
            acc0 = (_Float16)((float)acc0 + (float)(_Float16)t0);
            acc1 = (_Float16)((float)acc1 + (float)(_Float16)t1);
        }
    }
    iris_h2v o; o.x = acc0; o.y = acc1;
    a.feat[(size_t)level * a.n_chunk + i] = __builtin_bit_cast(uint32_t, o);
}

__device__ __forceinline__ float ngp_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// The reference's output stage (model/brdf.py:255): tiny-cuda-nn hands back HALF, `.sigmoid()` is taken on that half tensor (torch: f32 arithmetic, ONE rounding
// to half) and only then `.float()` -- every albedo / metallic value, and the roughness before `* 0.98 + 0.02`, lies on the half grid.
__device__ __forceinline__ float ngp_out(float acc) { return (float)(_Float16)ngp_sigmoid((float)(_Float16)acc); }

// relu + f32 -> f16 of 8 accumulator registers: the B fragment of the next layer's k-step
__device__ __forceinline__ iris_h8 ngp_pack_relu(const iris_f16v& acc, int s) {
    iris_h8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = acc[8 * s + j]; r[j] = (_Float16)(v > 0.f ? v : 0.f); }
    return r;
}
// A fragment of a layer whose B operand is the previous layer's accumulator tile `b`, k-step s: element j is input neuron
// 32 b + 16 s + 8 (j >> 2) + 4 h + (j & 3)  (the row an accumulator register holds: row = (reg & 3) + 8 (reg >> 2) + 4 h)
__device__ __forceinline__ iris_h8 ngp_load_a_perm(const _Float16* W, int row, int b, int s, int h, bool valid) {
    iris_h8 r;
    const int k0 = 32 * b + 16 * s + 4 * h;
    const iris_h4 lo = valid ? *reinterpret_cast<const iris_h4*>(W + row * kNgpWidth + k0) : iris_h4{0, 0, 0, 0};
    const iris_h4 hi = valid ? *reinterpret_cast<const iris_h4*>(W + row * kNgpWidth + k0 + 8) : iris_h4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
    return r;
}

// One wave = 32 points per trip; 4 waves per workgroup, persistent over the chunk's 32-point tiles.
__global__ __launch_bounds__(256) void ngp_mlp_kernel(NgpArgs a) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const _Float16* W1 = a.w;
    const _Float16* W2 = a.w + kNgpWidth * 64;
    const _Float16* W3 = W2 + kNgpWidth * kNgpWidth;
    // weights as A fragments, resident for the life of the wave
    iris_h8 a1[2][4], a2[2][2][2], a3[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) a1[b][s] = *reinterpret_cast<const iris_h8*>(W1 + (32 * b + r) * 64 + 16 * s + 8 * h);     // natural k order: k = 16 s + 8 h + j
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s) a2[b2][b][s] = ngp_load_a_perm(W2, 32 * b2 + r, b, s, h, true);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) a3[b][s] = ngp_load_a_perm(W3, r, b, s, h, r < kNgpOutPad);                               // rows 16 .. 31 of the 32-row tile are zero

    const int n_tiles = (a.n + 31) >> 5;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    for (int t = wave; t < n_tiles; t += n_waves) {
        const int pt = t * 32 + r;
        const bool pv = pt < a.n;
        // layer 1: B fragment of k-step s = features 16 s + 8 h .. + 7 of point pt = levels 8 s + 4 h .. + 3 (two features each)
        iris_f16v acc1[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc1[b][q] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint32_t f[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = pv ? a.feat[(size_t)(8 * s + 4 * h + q) * a.n_chunk + pt] : 0u;
            iris_h8 bf;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const iris_h2v v = __builtin_bit_cast(iris_h2v, f[q]); bf[2 * q] = v.x; bf[2 * q + 1] = v.y; }
#pragma unroll
            for (int b = 0; b < 2; ++b) acc1[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[b][s], bf, acc1[b], 0, 0, 0);
        }
        // layer 2: relu(H1) straight from the accumulators
        iris_f16v acc2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc2[b][q] = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const iris_h8 bf = ngp_pack_relu(acc1[b], s);
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) acc2[b2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[b2][b][s], bf, acc2[b2], 0, 0, 0);
            }
        // output layer (no activation)
        iris_f16v acc3;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc3[q] = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3[b][s], ngp_pack_relu(acc2[b], s), acc3, 0, 0, 0);
        // accumulator row = (reg & 3) + 8 (reg >> 2) + 4 h: lanes h = 0 hold outputs 0 .. 3 in registers 0 .. 3, lanes h = 1 output 4 in register 0
        if (pv) {
            const int64_t g = a.n0 + pt;
            if (h == 0) {
                // model/brdf.py:255-260: half -> sigmoid -> half -> float; albedo = [..., :3], roughness = [..., 3:4] * 0.98 + 0.02
                a.albedo[g * 3] = ngp_out(acc3[0]); a.albedo[g * 3 + 1] = ngp_out(acc3[1]); a.albedo[g * 3 + 2] = ngp_out(acc3[2]);
                a.rough[g] = ngp_out(acc3[3]) * 0.98f + 0.02f;          // (in f32, after .float(): model/brdf.py:258)
            } else {
                a.metal[g] = ngp_out(acc3[0]);
            }
        }
    }
}

}  // namespace iris
