"""ORACLE (test infrastructure, never imported by the product): a pure-torch restatement of the material network the reference
runs its refine / emitter-training stages through -- `NGPBRDF.forward` (/root/reference/model/brdf.py:213-260): a tiny-cuda-nn
`NetworkWithInputEncoding(3, 5, HashGrid{32 levels x 2 features, 2^19 entries, base 16, x1.3}, FullyFusedMLP{64 x 2 hidden, ReLU})`
followed by a sigmoid.

PARITY UNPINNED.  tiny-cuda-nn (git submodule of no version pinned in /root/reference; `import tinycudann as tcnn`, model/brdf.py:10) is not
in this image and cannot be built here (CUDA only), the reference ships no checkpoint and no test vector for it.  What is restated is the
library's PUBLISHED algorithm (Mueller et al., "Instant Neural Graphics Primitives with a Multiresolution Hash Encoding", SIGGRAPH 2022,
section 3; tiny-cuda-nn's `grid.h` / `fully_fused_mlp.cu` as documented there):

  * level l: scale = 2^(l * log2(1.3)) * 16 - 1, resolution = ceil(scale) + 1, table entries = min(round_up(resolution^3, 8), 2^19), 2 features each
  * a position x in the encoding's input space: p = x * scale + 0.5, cell = floor(p) (as uint32: negative cells wrap), w = p - cell;
    the 8 corners are indexed densely (x + y * res + z * res^2) while that fits the table, otherwise by the coherent prime hash
    (x * 1) ^ (y * 2654435761) ^ (z * 805459861), both modulo the table size; features are interpolated trilinearly, accumulating in half
  * parameters: ONE flat float32 tensor `mlp.params` = [MLP weights | grid tables]; the MLP weights are row-major (out x in) matrices
    64x64, 64x64, 16x64 (5 outputs padded to 16), computed in half
  * the reference feeds `position * 2 - 1` with position normalised to [0, 1] (model/brdf.py:252-254), i.e. inputs in [-1, 1]; the library
    does not clamp, negative cells wrap around in uint32 -- restated as is
  * outputs (model/brdf.py:255-260): the library hands back HALF; `.sigmoid()` runs on that half tensor (torch: f32 arithmetic, one rounding to half);
    `.float()` afterwards -> albedo (3), roughness * 0.98 + 0.02 (in f32), metallic.  Restated as is: accumulators -> half -> sigmoid -> half -> float,
    so every albedo / metallic value (and the roughness before its affine map) lies on the half grid, as the reference's do

The HIP path (iris_amd/csrc/iris_ngp.h) is compared with THIS restatement (tests/test_ngp.py); it accumulates the MLP in f32 on the matrix cores
where tiny-cuda-nn's fully fused kernel accumulates in half, so the bar is a tolerance (written in the test), not bit-exactness."""
import math

import numpy as np
import torch

N_LEVELS, N_FEATURES, LOG2_HASHMAP, BASE_RES, PER_LEVEL_SCALE = 32, 2, 19, 16, 1.3
WIDTH, N_OUT, N_OUT_PADDED = 64, 5, 16
N_MLP_PARAMS = WIDTH * (N_LEVELS * N_FEATURES) + WIDTH * WIDTH + N_OUT_PADDED * WIDTH


def level_tables():
    """[(scale, resolution, entries, offset in entries)] per level + total entries (tiny-cuda-nn GridEncoding constructor)."""
    # (the library evaluates the scale in float with the device's fast exp2f, whose last bits are not reproducible; here -- and in iris_hip.hip -- every
    #  transcendental is taken in double and rounded to float once: log2(1.3f) -> float, level * that in float, exp2 -> float, * 16 - 1 in float)
    log2_scale = np.float32(np.log2(np.float64(np.float32(PER_LEVEL_SCALE))))
    rows, offset = [], 0
    for l in range(N_LEVELS):
        e = np.float32(np.float32(l) * log2_scale)
        scale = np.float32(np.float32(np.exp2(np.float64(e))) * np.float32(BASE_RES) - np.float32(1.0))
        res = int(math.ceil(float(scale))) + 1
        n = min(res ** 3, (2 ** 32 - 1) // 2)
        n = (n + 7) // 8 * 8
        n = min(n, 1 << LOG2_HASHMAP)
        rows.append((float(scale), res, n, offset))
        offset += n
    return rows, offset


def n_params():
    return N_MLP_PARAMS + level_tables()[1] * N_FEATURES


def encode(params, x):
    """HashGrid encoding of x (B,3) float32 in the encoding's input space -> (B, 64) float16.  params: flat float32 tensor (the grid part is read as half)."""
    rows, total = level_tables()
    grid = params[N_MLP_PARAMS:].to(torch.float16).reshape(total, N_FEATURES)
    B = x.shape[0]
    out = torch.empty(B, N_LEVELS * N_FEATURES, dtype=torch.float16, device=x.device)
    M32 = 0xFFFFFFFF
    for l, (scale, res, n, off) in enumerate(rows):
        p = torch.from_numpy(_fma(x.cpu().numpy().astype(np.float32), np.float32(scale), np.float32(0.5))).to(x.device)      # fmaf(scale, x, 0.5)
        fl = torch.floor(p)
        w = p - fl                                                              # in [0, 1)
        cell = fl.to(torch.int64) & M32                                         # (uint32_t)(int)floor: negative cells wrap
        acc = torch.zeros(B, N_FEATURES, dtype=torch.float16, device=x.device)
        for corner in range(8):
            wgt = torch.ones(B, dtype=torch.float32, device=x.device)
            c = []
            for d in range(3):
                if corner & (1 << d):
                    wgt = wgt * w[:, d]; c.append((cell[:, d] + 1) & M32)
                else:
                    wgt = wgt * (np.float32(1.0) - w[:, d]); c.append(cell[:, d])
            stride, index, dims_used = 1, torch.zeros(B, dtype=torch.int64, device=x.device), 0
            for d in range(3):
                if stride > n:
                    break
                index = (index + c[d] * stride) & M32
                stride *= res
                dims_used += 1
            if n < stride:                                                      # the table is smaller than the dense grid: hash
                index = (c[0] * 1) & M32
                index = index ^ ((c[1] * 2654435761) & M32)
                index = index ^ ((c[2] * 805459861) & M32)
            index = index % n
            val = grid[off + index].to(torch.float32)                           # (B, 2)
            acc = (acc.to(torch.float32) + (wgt[:, None] * val).to(torch.float16).to(torch.float32)).to(torch.float16)    # result += (half)(weight * value), a half add
        out[:, l * N_FEATURES:(l + 1) * N_FEATURES] = acc
    return out


def _fma(a, b, c):
    """fmaf(a, b, c) for float32 arrays (the product of two float32 is exact in float64; one rounding to float32)."""
    return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(np.float32)


def mlp(params, feat, accumulate=torch.float32):
    """FullyFusedMLP{64, 2 hidden layers, ReLU, no output activation}: half weights and activations; `accumulate` = the dtype of the dot products
    (float32: the matrix cores of the HIP path; the library's fused kernel uses half accumulators)."""
    w = params[:N_MLP_PARAMS].to(torch.float16)
    n_in = N_LEVELS * N_FEATURES
    W1 = w[:WIDTH * n_in].reshape(WIDTH, n_in)
    W2 = w[WIDTH * n_in:WIDTH * n_in + WIDTH * WIDTH].reshape(WIDTH, WIDTH)
    W3 = w[WIDTH * n_in + WIDTH * WIDTH:].reshape(N_OUT_PADDED, WIDTH)
    h = feat.to(torch.float16)
    for W in (W1, W2):
        h = torch.relu((h.to(accumulate) @ W.to(accumulate).T)).to(torch.float16)
    return (h.to(accumulate) @ W3.to(accumulate).T)[:, :N_OUT]                # (B, 5) in `accumulate`


def forward(params, position, voxel_min, voxel_max):
    """NGPBRDF.forward (model/brdf.py:243-260): position (B,3) world space -> {'albedo' (B,3), 'roughness' (B,1), 'metallic' (B,1)} float32."""
    pos = (position.to(torch.float32) - np.float32(voxel_min)) / np.float32(float(voxel_max) - float(voxel_min))       # python floats: the difference is taken in double
    x = pos * np.float32(2.0) - np.float32(1.0)
    out_half = mlp(params, encode(params, x)).to(torch.float16)                                     # what tcnn returns
    mat = torch.sigmoid(out_half.to(torch.float32)).to(torch.float16).to(torch.float32)              # half sigmoid = f32 arithmetic + one rounding; then .float()
    return {"albedo": mat[:, :3].contiguous(), "roughness": mat[:, 3:4] * np.float32(0.98) + np.float32(0.02), "metallic": mat[:, 4:5].contiguous()}
