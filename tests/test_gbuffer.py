"""pooling builders (VoxelSLF.scatter_add, voxel occupancy histogram, per-triangle sums).
Oracle = numpy restatements citing the reference lines; VoxelSLF.scatter_add is pinned by a golden from the reference class."""
import numpy as np
import pytest
import torch

from conftest import golden


def np_scatter_add(inds, H, vmin, vmax, x, rgb, kv):
    """model/slf.py:41-61 in numpy: spatial_idx, then sequential scatter_add of radiance and count."""
    q = ((x - np.float32(vmin)) / np.float32(vmax - vmin) * np.float32(H)).astype(np.int64).clip(0, H - 1)
    idx = inds[q[:, 2], q[:, 1], q[:, 0]]
    rad = np.zeros((kv, 3), np.float32); cnt = np.zeros(kv, np.int64)
    np.add.at(rad, idx, rgb); np.add.at(cnt, idx, 1)
    return rad, cnt


def np_voxel_histogram(x, vmin, vmax, H):
    """slf_bake.py:104-110"""
    q = ((x - np.float32(vmin)) / np.float32(vmax - vmin) * np.float32(H)).astype(np.int64).clip(0, H - 1)
    lin = q[:, 0] + q[:, 1] * H + q[:, 2] * H * H
    return np.bincount(lin, minlength=H ** 3).astype(np.float32).reshape(H, H, H)


def test_numpy_restatement_matches_reference_scatter_add():
    g = golden("slf_scatter.npz")
    kk, jj, ii = np.where(g["mask"])
    inds = -np.ones(g["mask"].shape, np.int64); inds[kk, jj, ii] = np.arange(len(ii))
    rad, cnt = np_scatter_add(inds, g["mask"].shape[0], float(g["voxel_min"]), float(g["voxel_max"]), g["x"], g["rgb"], len(ii))
    np.testing.assert_array_equal(cnt, g["count"])
    np.testing.assert_allclose(rad, g["radiance"], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_hip_pooling_builders():
    from iris_amd.model.slf import VoxelSLF
    from iris_amd.utils.gbuffer import voxel_histogram, scatter_add_rows
    dev = torch.device("cuda:0")
    g = golden("slf_scatter.npz")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    slf = VoxelSLF(torch.from_numpy(g["mask"]), float(g["voxel_min"]), float(g["voxel_max"])).to(dev)
    slf.scatter_add(T(g["x"]), T(g["rgb"]))
    slf.scatter_add(T(g["x"][:0]), T(g["rgb"][:0]))                     # empty batch
    np.testing.assert_array_equal(slf.count.cpu().numpy(), g["count"])   # integer work: exact
    np.testing.assert_allclose(slf.radiance.cpu().numpy(), g["radiance"], rtol=1e-5, atol=1e-5)   # float sums: order only
    # mean pooling as slf_bake.py:138, then the lookup sees the new table
    slf.radiance = slf.radiance / slf.count[..., None].float().clamp_min(1)
    slf.refresh()
    rgb = slf(T(g["x"][:64]))["rgb"].cpu().numpy()
    assert np.isfinite(rgb).all() and rgb.max() <= 3.0 + 1e-5
    # occupancy histogram
    H = 32
    x = (np.random.default_rng(0).random((50000, 3)).astype(np.float32) * 4.4 - 0.2)
    hist = voxel_histogram(T(x), -0.1, 4.1, H)
    np.testing.assert_array_equal(hist.cpu().numpy(), np_voxel_histogram(x, -0.1, 4.1, H))
    # per-triangle sums
    rng = np.random.default_rng(1)
    idx = rng.integers(-1, 100, size=20000).astype(np.int64); vals = rng.random((20000, 3)).astype(np.float32)
    out = torch.zeros(100, 3, device=dev); cnt = torch.zeros(100, device=dev)
    scatter_add_rows(T(vals), T(idx), out, cnt)
    ref = np.zeros((100, 3), np.float32); rc = np.zeros(100, np.float32); m = idx >= 0
    np.add.at(ref, idx[m], vals[m]); np.add.at(rc, idx[m], 1)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
