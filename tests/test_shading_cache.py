"""8(f)-3: packed shading cache + the BRDF trainer's shading combine (utils/dataset/scannetpp/dataset.py:359-377,409-414;
train_brdf_crf.py:195-203).  GPU tests call the HIP kernels through the C ABI and compare with the golden vectors captured from the
reference (tests/golden/shade_cached.npz) and bit for bit with the oracle."""
import os

import numpy as np
import pytest

from conftest import golden, rel_l2

R = 6


def _maps(g):
    return g["map_0"], [g[f"map_{1 + j}"] for j in range(R)], [g[f"map_{1 + R + j}"] for j in range(R)]


@pytest.mark.gpu
def test_pack_and_slice_match_reference():
    import torch
    from iris_amd.utils.shading_cache import ShadingCache
    g = golden("shade_cached.npz")
    d, s0, s1 = _maps(g)
    dev = torch.device("cuda:0")
    cache = ShadingCache(len(d), R, dev)
    assert cache.row_floats == 40
    cache.put_view(0, torch.from_numpy(d).to(dev), [torch.from_numpy(m).to(dev) for m in s0], [torch.from_numpy(m).to(dev) for m in s1])
    idx = torch.from_numpy(g["idx"]).to(dev)
    di, sp0, sp1 = cache.gather(idx)
    np.testing.assert_array_equal(di.cpu().numpy(), g["diffuse"])
    np.testing.assert_array_equal(sp0.cpu().numpy(), g["specular0"])
    np.testing.assert_array_equal(sp1.cpu().numpy(), g["specular1"])
    full = torch.cat([t.reshape(len(d), -1) for t in cache.gather(None)], 1)
    np.testing.assert_array_equal(full.cpu().numpy(), g["all_cache"])
    rows = cache.rows.cpu().numpy()
    np.testing.assert_array_equal(rows[:, 3], 0)                       # pad lane
    np.testing.assert_array_equal(rows[:, 4 + 6 * 2:4 + 6 * 2 + 3], s0[2])   # level-interleaved: level 2 spec0, then spec1
    np.testing.assert_array_equal(rows[:, 4 + 6 * 2 + 3:4 + 6 * 3], s1[2])


@pytest.mark.gpu
def test_shade_cached_forward_backward(oracle_mod):
    import torch
    from iris_amd.utils.shading_cache import ShadingCache
    g = golden("shade_cached.npz")
    d, s0, s1 = _maps(g)
    dev = torch.device("cuda:0")
    cache = ShadingCache(len(d), R, dev)
    cache.put_view(0, torch.from_numpy(d).to(dev), [torch.from_numpy(m).to(dev) for m in s0], [torch.from_numpy(m).to(dev) for m in s1])
    idx = torch.from_numpy(g["idx"]).to(dev)
    albedo = torch.from_numpy(g["albedo"]).to(dev).requires_grad_(True)
    metallic = torch.from_numpy(g["metallic"]).to(dev).requires_grad_(True)
    rough = torch.from_numpy(g["roughness"]).to(dev).requires_grad_(True)
    Lc = cache.shade(idx, albedo, metallic, rough)
    np.testing.assert_array_equal(Lc.detach().cpu().numpy(), g["L"])            # forward: bit-exact with the reference's torch ops
    ga, gm, gr = torch.autograd.grad(Lc, [albedo, metallic, rough], torch.from_numpy(g["gL"]).to(dev))
    assert gm.shape == metallic.shape and gr.shape == rough.shape
    # reference autograd (its own summation order): north_star tolerance 1e-4 rel-L2, observed ~1e-7
    for mine, ref in ((ga, g["g_albedo"]), (gm, g["g_metallic"]), (gr, g["g_roughness"])):
        assert rel_l2(mine.cpu().numpy(), ref) < 1e-6
    # oracle (same summation order): bit-exact
    _, oa, om, og = oracle_mod.shade_cached(g["all_cache"], g["idx"], g["albedo"], g["metallic"], g["roughness"], g["gL"])
    np.testing.assert_array_equal(ga.cpu().numpy(), oa)
    np.testing.assert_array_equal(gm.cpu().numpy(), om)
    np.testing.assert_array_equal(gr.cpu().numpy(), og)
    # only the requested gradients are produced
    Lc2 = cache.shade(idx, albedo.detach(), metallic.detach(), rough)
    (gr2,) = torch.autograd.grad(Lc2, [rough], torch.from_numpy(g["gL"]).to(dev))
    np.testing.assert_array_equal(gr2.cpu().numpy(), og)


@pytest.mark.gpu
def test_shade_cached_large_random_vs_oracle(oracle_mod):
    """2 M rows, random permutation batch, odd level count (padded rows), empty batch."""
    import torch
    from iris_amd.utils.shading_cache import ShadingCache
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    for Rl, n, B in ((6, 200_000, 65_536), (5, 10_000, 4096), (1, 1000, 1000)):
        maps = [rng.random((n, 3), dtype=np.float32) * 3 for _ in range(1 + 2 * Rl)]
        cache = ShadingCache(n, Rl, dev)
        assert cache.row_floats % 4 == 0
        cache.put_view(0, torch.from_numpy(maps[0]).to(dev), [torch.from_numpy(m).to(dev) for m in maps[1:1 + Rl]],
                       [torch.from_numpy(m).to(dev) for m in maps[1 + Rl:]])
        rows_ref = oracle_mod.cache_pack(maps[0], maps[1:1 + Rl], maps[1 + Rl:])
        idx = rng.permutation(n)[:B].astype(np.int64)
        albedo = rng.random((B, 3), dtype=np.float32); metallic = rng.random((B, 1), dtype=np.float32)
        rough = (rng.random((B, 1), dtype=np.float32) * 0.98 + 0.02).astype(np.float32) if Rl > 1 else np.full((B, 1), 0.02, np.float32)
        gL = rng.standard_normal((B, 3)).astype(np.float32)
        a = torch.from_numpy(albedo).to(dev).requires_grad_(True); m = torch.from_numpy(metallic).to(dev).requires_grad_(True)
        r = torch.from_numpy(rough).to(dev).requires_grad_(True)
        Lc = cache.shade(torch.from_numpy(idx).to(dev), a, m, r)
        ga, gm, gr = torch.autograd.grad(Lc, [a, m, r], torch.from_numpy(gL).to(dev))
        oL, oa, om, og = oracle_mod.shade_cached(rows_ref, idx, albedo, metallic, rough, gL)
        np.testing.assert_array_equal(Lc.detach().cpu().numpy(), oL)
        np.testing.assert_array_equal(ga.cpu().numpy(), oa)
        np.testing.assert_array_equal(gm.cpu().numpy(), om)
        np.testing.assert_array_equal(gr.cpu().numpy(), og)
    e = cache.shade(torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, 3, device=dev), torch.empty(0, 1, device=dev),
                    torch.empty(0, 1, device=dev))
    assert e.shape == (0, 3)


@pytest.mark.gpu
def test_cache_from_bake_and_exr_round_trip(tmp_path):
    """bake_view -> table (no files) equals bake_view -> EXR files -> table."""
    import torch
    from iris_amd import bake_shading as bs
    from iris_amd.utils import exr
    from iris_amd.utils.dataset import real_ldr
    from iris_amd.utils.shading_cache import ShadingCache
    from tools import synth
    from test_sharding_gpu import _setup
    dev = torch.device("cuda:0")
    os.makedirs(tmp_path / "em", exist_ok=True)
    g, scene, emitter = _setup(dev, str(tmp_path / "em"))
    H, W = 24, 32
    views = []
    for v in range(2):
        K, _ = synth.camera(H, W, v)
        xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), g["c2w"], False, device=dev)
        views.append(bs.bake_view(scene, emitter, xs, ds, 16, [8] * 6, seed=v, image_width=W))
    c1 = ShadingCache.from_bake(views)
    for i, v in enumerate(views):
        os.makedirs(tmp_path / "diffuse", exist_ok=True); os.makedirs(tmp_path / "specular", exist_ok=True)
        exr.write_exr(str(tmp_path / "diffuse" / ("%03d.exr" % i)), v["diffuse"].reshape(H, W, 3).cpu().numpy())
        for j in range(6):
            exr.write_exr(str(tmp_path / "specular" / ("%03d_0_%d.exr" % (i, j))), v["specular0"][j].reshape(H, W, 3).cpu().numpy())
            exr.write_exr(str(tmp_path / "specular" / ("%03d_1_%d.exr" % (i, j))), v["specular1"][j].reshape(H, W, 3).cpu().numpy())
    c2 = ShadingCache.from_exr_dir(str(tmp_path), 2, 6, dev)
    assert torch.equal(c1.rows, c2.rows)
    assert float(c1.rows.abs().sum()) > 0
