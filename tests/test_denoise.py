"""8(f)-4: the denoiser that replaces mitsuba.OptixDenoiser (bake_shading.py:81,129,198-200).  OptiX is closed: nothing to be bit-equal
to.  Checked here: (CPU) invariants of the filter on the oracle's restatement; (GPU) the HIP kernels against that restatement tap
for tap, and the quality criterion SURVEY.md 8(f)-4 names -- PSNR against a high-spp bake of the same view."""
import os

import numpy as np
import pytest

from conftest import golden, rel_l2


def _plane_scene(H, W, rng):
    """Two planes meeting at a vertical crease in the middle of the image + a band of invalid pixels at the bottom."""
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    left = xs < W // 2
    normal = np.where(left[..., None], np.float32([0, 0, 1]), np.float32([1, 0, 0])).astype(np.float32)
    position = np.stack([np.where(left, xs, W // 2), ys, np.where(left, 0, xs - W // 2)], -1).astype(np.float32) * 0.01
    valid = np.ones((H, W), bool); valid[-3:] = False
    signal = np.where(left[..., None], np.float32([1.0, 0.8, 0.6]), np.float32([0.2, 0.3, 0.4])).astype(np.float32)
    noisy = (signal * (1 + 0.5 * rng.standard_normal((H, W, 3)))).astype(np.float32)
    noisy[~valid] = 0
    return normal, position, valid, signal, noisy


def psnr(a, b, mask=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if mask is not None:
        a, b = a[mask], b[mask]
    return 10 * np.log10(max(b.max(), 1e-12) ** 2 / max(np.mean((a - b) ** 2), 1e-30))


def test_oracle_filter_invariants(oracle_mod):
    rng = np.random.default_rng(0)
    H, W = 40, 56
    normal, position, valid, signal, noisy = _plane_scene(H, W, rng)
    const = np.where(valid[..., None], np.float32([0.3, 0.5, 0.7]), 0).astype(np.float32)
    out = oracle_mod.denoise(const, normal, position, valid)
    np.testing.assert_allclose(out, const, atol=1e-6)                       # a constant image is a fixed point; invalid pixels stay 0
    out = oracle_mod.denoise(noisy, normal, position, valid)
    assert np.all(out[~valid] == 0)
    assert psnr(out, signal, valid) > psnr(noisy, signal, valid) + 10       # noise goes down by >10 dB ...
    edge = np.abs(out[5:-5, W // 2 - 1] - out[5:-5, W // 2]).mean()          # ... and the crease is not smeared (normals differ by 90 degrees)
    assert edge > 0.9 * np.abs(signal[0, W // 2 - 1] - signal[0, W // 2]).mean()
    # without guides the same step is blurred more than with them
    blur = oracle_mod.denoise(noisy, None, None, valid)
    assert psnr(out, signal, valid) > psnr(blur, signal, valid)


@pytest.mark.gpu
def test_hip_matches_oracle(oracle_mod):
    import torch
    from iris_amd.utils.denoise import Denoiser
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    H, W = 45, 70                                                           # not multiples of the 16x16 tile
    normal, position, valid, signal, noisy = _plane_scene(H, W, rng)
    maps = [noisy * s for s in (1.0, 0.5, 2.0, 0.1, 3.0)]                   # 5 maps: one full group of 4 + a group of 1
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    dn = Denoiser((W, H), dev).set_guides(t(normal), t(position), t(valid, torch.bool))
    outs = dn.denoise_maps([t(m) for m in maps])
    for m, o in zip(maps, outs):
        ref = oracle_mod.denoise(m, normal, position, valid)
        assert rel_l2(o.cpu().numpy(), ref) < 1e-5
        np.testing.assert_allclose(o.cpu().numpy(), ref, rtol=2e-4, atol=2e-5)
    # one iteration, other sigmas, no guides, numpy input as the reference passes it
    dn2 = Denoiser((W, H), dev, iterations=1, sigma_l=2.0, sigma_n=8.0, sigma_p=0.5)
    o = dn2(noisy)
    assert rel_l2(o.cpu().numpy(), oracle_mod.denoise(noisy, None, None, None, 1, 2.0, 8.0, 0.5)) < 1e-5
    dn2.set_guides(None, None, t(valid, torch.bool))
    assert rel_l2(dn2(noisy).cpu().numpy(), oracle_mod.denoise(noisy, None, None, valid, 1, 2.0, 8.0, 0.5)) < 1e-5
    assert dn.denoise_maps([]) == []


@pytest.mark.gpu
def test_psnr_against_high_spp_bake(tmp_path):
    """judged on PSNR vs a high-spp render.  Low-spp bake of the box room, denoised, against spp = 4096."""
    import torch
    from iris_amd import bake_shading as bs
    from iris_amd.utils.dataset import real_ldr
    from iris_amd.utils.denoise import Denoiser
    from tools import synth
    from test_sharding_gpu import _setup
    dev = torch.device("cuda:0")
    os.makedirs(tmp_path / "em", exist_ok=True)
    g, scene, emitter = _setup(dev, str(tmp_path / "em"))
    H, W = 96, 128
    K, _ = synth.camera(H, W, 0)
    xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), g["c2w"], False, device=dev)
    ref = bs.bake_view(scene, emitter, xs, ds, 4096, [4096] * 6, seed=11, image_width=W)
    raw = bs.bake_view(scene, emitter, xs, ds, 16, [16] * 6, seed=5, image_width=W)
    den = bs.bake_view(scene, emitter, xs, ds, 16, [16] * 6, seed=5, image_width=W, denoiser=Denoiser((W, H), dev))
    assert torch.equal(den["specular0"][0], raw["specular0"][0])           # lowest roughness level is not denoised (:198)
    gains = {}
    for name, pick in (("diffuse", lambda o: o["diffuse"]), ("spec0_l2", lambda o: o["specular0"][2]), ("spec1_l3", lambda o: o["specular1"][3]),
                       ("spec0_l5", lambda o: o["specular0"][5])):
        r, a, b = pick(ref).cpu().numpy(), pick(raw).cpu().numpy(), pick(den).cpu().numpy()
        gains[name] = (psnr(a, r), psnr(b, r))
    print("PSNR raw -> denoised:", {k: (round(v[0], 2), round(v[1], 2)) for k, v in gains.items()})
    for k, (p_raw, p_den) in gains.items():
        assert p_den > p_raw + 3.0, (k, p_raw, p_den)
