"""The hash-grid material network (NGPBRDF, /root/reference/model/brdf.py:213-260): tiny-cuda-nn is absent, so the oracle is the pure-torch
restatement of its published algorithm (oracle/ngp_torch.py, "parity unpinned").

  not gpu: the restatement against an independent scalar Python evaluation of the same published formulas (a few points, every level), the
           level table / parameter count, the state-dict key of the reference's checkpoints
  gpu:     the HIP path (gathers + MFMA perceptron) against the restatement: the encoded features feed a half-precision network whose accumulation
           order differs (f32 matrix-core accumulation against torch's f32 matmul), so the bar is a tolerance: |d| <= 4e-3 on every output, mean <= 3e-4;
           with weights chosen so that every product is exact in half and f32 (small integers) the two must agree to 3e-7 (one half ulp where the two
           exp() straddle a rounding boundary of the HALF output stage); every output lies on the half grid, as the reference's do (model/brdf.py:255)"""
import math
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

sys.path.insert(0, REPO)
from oracle import ngp_torch as ng     # noqa: E402


def _scalar_encode(params, x):
    """independent restatement, one point at a time with Python ints (uint32 arithmetic by masking)"""
    rows, total = ng.level_tables()
    grid = params[ng.N_MLP_PARAMS:].to(torch.float16).reshape(total, 2)
    M = 0xFFFFFFFF
    out = np.zeros(64, np.float16)
    for l, (scale, res, n, off) in enumerate(rows):
        w, cell = [], []
        for d in range(3):
            p = np.float32(np.float64(np.float32(scale)) * np.float64(np.float32(x[d])) + 0.5)
            fl = math.floor(float(p))
            w.append(np.float32(p - np.float32(fl))); cell.append(int(fl) & M)
        acc = np.zeros(2, np.float16)
        for corner in range(8):
            wgt = np.float32(1.0); g = []
            for d in range(3):
                if corner >> d & 1:
                    wgt = np.float32(wgt * w[d]); g.append((cell[d] + 1) & M)
                else:
                    wgt = np.float32(wgt * np.float32(np.float32(1.0) - w[d])); g.append(cell[d])
            if res ** 3 > n or res ** 2 > n or res > n:      # some dimension's stride exceeds the table
                stride, idx, hashed = 1, 0, False
                for d in range(3):
                    if stride > n:
                        break
                    idx = (idx + g[d] * stride) & M; stride *= res
                if n < stride:
                    idx = (g[0] ^ ((g[1] * 2654435761) & M) ^ ((g[2] * 805459861) & M)) & M
            else:
                idx = (g[0] + g[1] * res + g[2] * res * res) & M
                if n < res ** 3:
                    idx = (g[0] ^ ((g[1] * 2654435761) & M) ^ ((g[2] * 805459861) & M)) & M
            idx %= n
            v = grid[off + idx].numpy().astype(np.float32)
            acc = (acc.astype(np.float32) + (wgt * v).astype(np.float16).astype(np.float32)).astype(np.float16)
        out[2 * l:2 * l + 2] = acc
    return out


def _params(seed, scale=0.5):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(ng.n_params(), generator=g) * 2 - 1) * scale


def test_level_table_and_parameter_count():
    rows, total = ng.level_tables()
    assert rows[0][:3] == (15.0, 16, 4096) and rows[1][1] == 21 and rows[1][2] == 9264          # 16^3; ceil(15 * 1.3 + 0.3) + 1 = 21, 21^3 = 9261 -> 9264
    assert all(r[2] == 1 << 19 for r in rows[7:]) and rows[6][2] < 1 << 19                       # level 7 on: the hashed 2^19-entry tables
    assert ng.N_MLP_PARAMS == 64 * 64 + 64 * 64 + 16 * 64
    assert ng.n_params() == ng.N_MLP_PARAMS + 2 * total == 27963328


def test_restatement_against_scalar_evaluation():
    params = _params(1)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(6, 3, generator=g) * 2 - 1                 # the reference feeds [-1, 1]: negative cells wrap
    x[0] = torch.tensor([-1.0, 0.0, 1.0]); x[1] = torch.tensor([0.999999, -0.999999, 0.5])
    enc = ng.encode(params, x).numpy()
    for i in range(x.shape[0]):
        np.testing.assert_array_equal(enc[i], _scalar_encode(params, x[i].numpy()))
    out = ng.forward(params, x * 1.5 + 0.25, -1.25, 1.75)    # (x * 1.5 + 0.25 maps back to [-1, 1] under voxel_min = -1.25, voxel_max = 1.75)
    assert out["albedo"].shape == (6, 3) and out["roughness"].shape == (6, 1) and out["metallic"].shape == (6, 1)
    assert float(out["roughness"].min()) >= 0.02 and float(out["roughness"].max()) <= 1.0
    for k in ("albedo", "metallic"):                          # model/brdf.py:255: sigmoid of a half tensor, then .float()
        assert torch.equal(out[k].to(torch.float16).to(torch.float32), out[k])


@pytest.mark.gpu
def test_state_dict_key_and_checkpoint_loading(tmp_path):
    from iris_amd.model.brdf import NGPBRDF, load_ngpbrdf
    net = NGPBRDF(-1.0, 2.0)
    assert list(net.state_dict().keys()) == ["mlp.params"] and net.state_dict()["mlp.params"].shape == (ng.n_params(),)
    params = _params(3)
    torch.save({"state_dict": {"material.mlp.params": params, "emitter.radiance": torch.zeros(3)}}, tmp_path / "last.ckpt")     # the reference's checkpoint layout (refine_shading.py:84-89)
    net = load_ngpbrdf(-1.0, 2.0, str(tmp_path / "last.ckpt"))
    assert torch.equal(net.mlp.params, params) and not any(p.requires_grad for p in net.parameters())
    with pytest.raises(RuntimeError):
        net.load_state_dict({"mlp.params": torch.zeros(5)})


@pytest.mark.gpu
def test_copies_and_streams_of_one_network():
    """deepcopy / pickle of a network that has already run (its native handle is a raw pointer: the copy must build its own, not share or double-free it),
    and forwards of ONE handle on two streams (the handle's feature buffer is shared: the library orders them on the device)"""
    import copy
    import pickle
    from iris_amd.model.brdf import NGPBRDF
    dev = torch.device("cuda:0")
    net = NGPBRDF(-2.0, 2.5)
    net.load_state_dict({"mlp.params": _params(6, scale=0.3)})
    pos = (torch.rand(300000, 3, generator=torch.Generator().manual_seed(3)) * 4.5 - 2.0).to(dev)
    pos2 = pos.flip(0).contiguous()
    ref, ref2 = net(pos), net(pos2)
    torch.cuda.synchronize()
    for other in (copy.deepcopy(net), pickle.loads(pickle.dumps(net))):
        assert other._h is None
        out = other(pos)
        assert all(torch.equal(out[k], ref[k]) for k in ref)
        del other
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        with torch.cuda.stream(s1):
            a = net(pos)
        with torch.cuda.stream(s2):
            b = net(pos2)
        torch.cuda.synchronize()
        assert all(torch.equal(a[k], ref[k]) for k in ref) and all(torch.equal(b[k], ref2[k]) for k in ref2)
    assert all(torch.equal(net(pos)[k], ref[k]) for k in ref)          # (the original's handle is still alive)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 33, 5000, (1 << 20) + 77])
def test_hip_forward_against_the_restatement(n):
    from iris_amd.model.brdf import NGPBRDF
    dev = torch.device("cuda:0")
    params = _params(4, scale=0.3)
    net = NGPBRDF(-2.0, 2.5)
    net.load_state_dict({"mlp.params": params})
    g = torch.Generator().manual_seed(n)
    pos = torch.rand(n, 3, generator=g) * 4.5 - 2.0
    out = net(pos.to(dev))
    torch.cuda.synchronize()
    m = min(n, 20000)                                         # (the restatement is slow: the first m points and, for the chunked case, the last ones)
    sel = torch.cat([torch.arange(m), torch.arange(max(n - 200, 0), n)]).unique()
    ref = ng.forward(params, pos[sel], -2.0, 2.5)
    for k in ("albedo", "roughness", "metallic"):
        a, b = out[k].cpu()[sel], ref[k]
        assert a.shape == b.shape
        d = (a - b).abs()
        assert float(d.max()) <= 4e-3 and (n < 1000 or float(d.mean()) <= 3e-4), (k, float(d.max()), float(d.mean()))
    assert out["albedo"].shape == (n, 3) and out["roughness"].shape == (n, 1)


@pytest.mark.gpu
def test_hip_forward_exact_with_dyadic_weights():
    """A network whose every dot product is EXACT in f32 whatever the summation order: the tables hold one constant per feature (so the encoded
    features are ~0.25 / ~-0.5 with 11-bit mantissas, the same bits on both sides), W1 in {-2..2}/8, W2 and W3 in {-1,0,1}/4: layer-1 sums are
    multiples of 2^-15 below 8, layer-2 sums multiples of 2^-17, layer-3 sums multiples of 2^-19 -- all within 24 bits.  The matrix-core path must
    then reproduce the restatement up to the last-bit difference of the two exp() implementations: a wrong lane / register mapping of an MFMA
    operand, or a wrong k permutation between layers, cannot hide behind the tolerance of the random-weight test."""
    from iris_amd.model.brdf import NGPBRDF
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    params = torch.zeros(ng.n_params())
    n1, n2 = 64 * 64, 64 * 64
    params[:n1] = torch.randint(-2, 3, (n1,), generator=g).float() / 8
    params[n1:n1 + n2] = torch.randint(-1, 2, (n2,), generator=g).float() / 4
    params[n1 + n2:ng.N_MLP_PARAMS] = torch.randint(-1, 2, (ng.N_MLP_PARAMS - n1 - n2,), generator=g).float() / 4
    grid = params[ng.N_MLP_PARAMS:].reshape(-1, 2)
    grid[:, 0] = 0.25; grid[:, 1] = -0.5
    net = NGPBRDF(0.0, 1.0)
    net.load_state_dict({"mlp.params": params})
    pos = torch.rand(4096, 3, generator=g)
    out = net(pos.to(dev))
    ref = ng.forward(params, pos, 0.0, 1.0)
    pre = ng.mlp(params, ng.encode(params, pos * 2 - 1))
    assert float(pre.abs().max()) > 0.5 and float(pre.std()) > 0.1          # (a network that does something)
    for k in ("albedo", "roughness", "metallic"):
        d = np.abs(out[k].cpu().numpy() - ref[k].numpy())
        # the output stage rounds sigmoid(x) to HALF (model/brdf.py:255): where the two exp() implementations' last-bit difference straddles a half rounding
        # boundary the outputs differ by exactly one half ulp (<= 2^-12 below 1) -- a handful of values; everything else is equal to the last f32 bit
        assert float(d.max()) <= 2.0 ** -12 + 1e-7 and int((d > 3e-7).sum()) <= max(2, d.size // 2000), (k, float(d.max()), int((d > 3e-7).sum()))


@pytest.mark.gpu
def test_hip_outputs_lie_on_the_half_grid():
    """model/brdf.py:255-260: `self.mlp(x).sigmoid()` is a HALF tensor before `.float()`: albedo and metallic are exactly representable in half, and so is
    (roughness - 0.02) / 0.98 up to the f32 rounding of the affine map (checked through the forward map: some half value h gives the same f32 roughness)."""
    from iris_amd.model.brdf import NGPBRDF
    dev = torch.device("cuda:0")
    net = NGPBRDF(-2.0, 2.5)
    net.load_state_dict({"mlp.params": _params(9, scale=0.3)})
    pos = torch.rand(50000, 3, generator=torch.Generator().manual_seed(1)) * 4.5 - 2.0
    out = net(pos.to(dev))
    for k in ("albedo", "metallic"):
        v = out[k].cpu()
        assert torch.equal(v.to(torch.float16).to(torch.float32), v), k
    r = out["roughness"].cpu()
    h = ((r - 0.02) / 0.98).to(torch.float16).to(torch.float32)           # the nearest half candidate
    assert torch.equal(h * np.float32(0.98) + np.float32(0.02), r)
    assert float(r.min()) >= 0.02 and float(r.max()) <= 1.0


@pytest.mark.gpu
def test_hip_encoding_bit_exact():
    """the hash-grid encoding alone (iris_debug_ngp_encode) against the restatement: same operation sequence, same level table -> the same half bits"""
    from iris_amd import _lib as L
    from iris_amd.model.brdf import NGPBRDF
    dev = torch.device("cuda:0")
    params = _params(5, scale=0.3)
    net = NGPBRDF(-2.0, 2.5)
    net.load_state_dict({"mlp.params": params})
    g = torch.Generator().manual_seed(11)
    n = 3000
    pos = torch.rand(n, 3, generator=g) * 4.5 - 2.0
    pos[0] = torch.tensor([-2.0, 2.5, 0.25]); pos[1] = torch.tensor([2.5, -2.0, -2.0])        # the box corners: x = -1 / +1 exactly
    feat = torch.zeros(32, n, dtype=torch.int32, device=dev)
    L.check(L.lib().iris_debug_ngp_encode(net._handle(dev), L.ptr(pos.to(dev)), n, L.ptr(feat), L.stream()))
    torch.cuda.synchronize()
    hip = feat.cpu().numpy().view(np.uint16).reshape(32, n, 2)
    x = ((pos - np.float32(-2.0)) / np.float32(4.5)) * np.float32(2.0) - np.float32(1.0)
    ref = ng.encode(params, x).numpy().view(np.uint16).reshape(n, 32, 2).transpose(1, 0, 2)
    np.testing.assert_array_equal(hip, ref)
