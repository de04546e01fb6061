"""the stages that write vslf.npz / emitter.pth (slf_bake.py:69-145, slf_refine.py:85-108,
extract_emitter_ldr.py:72-115) run on the device and produce files the bake loads.
  test_drivers_vs_reference_replay: against tests/golden/prebake_box.npz -- the loop bodies of those scripts replayed through the reference's
      own Python (VoxelSLF class, torch ops; tools/make_driver_goldens.py) on the rays and photographs stored in the fixture.
  test_slf_bake_refine_and_emitter_extraction: a second, independent numpy restatement over views built here + the files load and bake."""
import os

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _views(g, dev, H=48, W=64, n=3):
    from iris_amd.utils.dataset import real_ldr
    from tools import synth
    out = []
    for v in range(n):
        K, c2w = synth.camera(H, W, v, n_views=n)
        xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w if v else g["c2w"], False, device=dev)
        out.append({"rays": torch.cat([xs, ds], -1)})
    # one more "view" looking up at the ceiling light from the room centre
    gx, gy = torch.meshgrid(torch.linspace(0.2, 3.8, W), torch.linspace(0.2, 2.8, H), indexing="xy")
    tgt = torch.stack([gx, gy, torch.full_like(gx, 2.6)], -1).reshape(-1, 3)
    o = torch.tensor([2.0, 1.5, 1.0]).expand_as(tgt)
    out.append({"rays": torch.cat([o, torch.nn.functional.normalize(tgt - o, dim=-1)], -1).to(dev)})
    return out


def test_slf_bake_refine_and_emitter_extraction(tmp_path):
    from iris_amd import slf_bake as sb, bake_shading as bs
    from iris_amd.model.emitter import SLFEmitter
    from iris_amd.utils.path_tracing import Scene, ray_intersect
    dev = torch.device("cuda:0")
    g = golden("bake_box.npz")
    scene = Scene(g["verts"], g["faces"], device=dev)
    views = _views(g, dev)
    # linear radiance "photographed" in each view: bright on the light quad (triangles 12, 13), a smooth field elsewhere
    hits = []
    for b in views:
        pos, _, _, idx, valid = ray_intersect(scene, b["rays"][:, :3].contiguous(), b["rays"][:, 3:].contiguous())
        rgb = 0.3 + 0.2 * torch.sin(pos * 2.0)
        rgb[idx >= 12] = torch.tensor([10.0, 9.0, 8.0], device=dev)
        rgb[~valid] = 0
        b["rgbs"] = rgb
        hits.append((pos.cpu().numpy(), idx.cpu().numpy(), valid.cpu().numpy(), rgb.cpu().numpy()))
    assert all(h[2].all() for h in hits)                                   # closed room

    res = 24
    sd = sb.bake_slf(scene, views, res_spatial=res, dataset="scannetpp", device=dev)
    # ---- numpy restatement of slf_bake.py:69-138 over the same hits
    P = np.concatenate([h[0][h[2]] for h in hits]); RGB = np.concatenate([h[3][h[2]] for h in hits])
    vmin, vmax = np.float32(min(1000., P.min())), np.float32(max(0.0, P.max()))
    c = vmin + vmax                                                         # slf_bake.py:90 (not halved)
    vmin, vmax = c + (vmin - c) * np.float32(1.1), c + (vmax - c) * np.float32(1.1)
    assert sd["voxel_min"] == pytest.approx(float(vmin), rel=1e-6) and sd["voxel_max"] == pytest.approx(float(vmax), rel=1e-6)
    fmin, fmax = np.float32(sd["voxel_min"]), np.float32(sd["voxel_max"])
    q = ((P - fmin) / np.float32(float(sd["voxel_max"]) - float(sd["voxel_min"])) * np.float32(res)).astype(np.int64).clip(0, res - 1)
    lin = q[:, 0] + q[:, 1] * res + q[:, 2] * res * res
    mask = (np.bincount(lin, minlength=res ** 3) > 0).reshape(res, res, res)
    np.testing.assert_array_equal(sd["mask"].numpy(), mask)
    kk, jj, ii = np.where(mask)
    inds = -np.ones(mask.shape, np.int64); inds[kk, jj, ii] = np.arange(len(ii))
    row = inds[q[:, 2], q[:, 1], q[:, 0]]
    assert (row >= 0).all()
    rad = np.zeros((len(ii), 3), np.float64); cnt = np.zeros(len(ii), np.int64)
    np.add.at(rad, row, RGB.astype(np.float64)); np.add.at(cnt, row, 1)
    np.testing.assert_array_equal(sd["weight"]["count"].numpy(), cnt)
    np.testing.assert_array_equal(sd["weight"]["inds"].numpy(), inds)
    np.testing.assert_allclose(sd["weight"]["radiance"].numpy(), rad / np.maximum(cnt, 1)[:, None], rtol=2e-5, atol=1e-6)

    # synthetic / real datasets scale the bounds about the origin (slf_bake.py:86-88)
    lo, hi = sb.scene_bounds(scene, views, "synthetic", dev)
    assert float(lo) == pytest.approx(1.1 * float(min(1000., P.min())), rel=1e-6) and float(hi) == pytest.approx(1.1 * float(P.max()), rel=1e-6)

    # ---- slf_refine: same grid, doubled radiance -> doubled means, counts restart
    for b in views:
        b["rgbs"] = b["rgbs"] * 2
    sd2 = sb.refine_slf(sd, scene, views, dev)
    np.testing.assert_array_equal(sd2["weight"]["count"].numpy(), cnt)
    np.testing.assert_allclose(sd2["weight"]["radiance"].numpy(), 2 * sd["weight"]["radiance"].numpy(), rtol=2e-5, atol=1e-6)

    # ---- extract_emitter_ldr: the two light triangles, their areas and normals
    em = sb.extract_emitters(scene, g["verts"], g["faces"], views, threshold=5.0, device=dev)
    np.testing.assert_array_equal(em["is_emitter"].numpy(), g["is_emitter"])
    np.testing.assert_allclose(em["emitter_area"].numpy(), g["emitter_area"], rtol=1e-6)
    assert em["emitter_vertices"].shape == (2, 3, 3) and em["emitter_radiance"].shape == (len(g["faces"]), 3)
    np.testing.assert_allclose(np.abs(em["emitter_normal"].numpy()), [[0, 0, 1], [0, 0, 1]], atol=1e-6)

    # ---- the files load where the reference's would, and the bake runs on them
    em["emitter_radiance"][:2] = torch.tensor([10.0, 9.0, 8.0])            # mode 'update' (:117-122): rows < K hold the learned radiance
    ep, sp = str(tmp_path / "emitter.pth"), str(tmp_path / "vslf.npz")
    torch.save(em, ep); torch.save(sd, sp)
    emitter = SLFEmitter(ep, sp)
    rays = views[0]["rays"]
    out = bs.bake_view(scene, emitter, rays[:, :3].contiguous(), rays[:, 3:].contiguous(), 16, [8] * 6, image_width=64)
    assert out["n_valid"] == rays.shape[0] and float(out["diffuse"].mean()) > 0.05


def test_drivers_vs_reference_replay():
    """bake_slf / refine_slf / extract_emitters against the reference's scripts replayed in its own Python (prebake_box.npz)"""
    from iris_amd import slf_bake as sb
    from iris_amd.utils.path_tracing import Scene
    dev = torch.device("cuda:0")
    g, f = golden("bake_box.npz"), golden("prebake_box.npz")
    scene = Scene(g["verts"], g["faces"], device=dev)
    views = [{"rays": torch.from_numpy(f[f"rays_{k}"]).to(dev), "rgbs": torch.from_numpy(f[f"rgbs_{k}"]).to(dev)} for k in range(int(f["n_views"]))]
    res = int(f["res_spatial"])
    # slf_bake.py:69-93: bounds (both dataset conventions)
    lo, hi = sb.scene_bounds(scene, views, "synthetic", dev)
    np.testing.assert_allclose([float(lo), float(hi)], f["bounds_synthetic"], rtol=1e-6, atol=1e-7)
    sd = sb.bake_slf(scene, views, res_spatial=res, dataset="scannetpp", device=dev)
    assert sd["voxel_min"] == pytest.approx(float(f["voxel_min"]), rel=1e-6, abs=1e-7) and sd["voxel_max"] == pytest.approx(float(f["voxel_max"]), rel=1e-6)
    # :95-114 occupancy; :116-138 pooling through the reference's VoxelSLF
    np.testing.assert_array_equal(sd["mask"].numpy(), f["mask"])
    np.testing.assert_array_equal(sd["weight"]["inds"].numpy(), f["slf_inds"])
    np.testing.assert_array_equal(sd["weight"]["count"].numpy(), f["slf_count"])
    np.testing.assert_allclose(sd["weight"]["radiance"].numpy(), f["slf_radiance"], rtol=2e-5, atol=1e-6)       # (float sums: order of the atomics)
    hist = sb.visible_voxels(scene, views, torch.tensor(sd["voxel_min"]), torch.tensor(sd["voxel_max"]), res, dev)
    np.testing.assert_array_equal(hist.cpu().numpy(), f["hist"])
    assert not any("_iris_hits" in b for b in views)                         # nothing stays pinned on the caller's batches
    # slf_refine.py:90-106
    for b in views:
        b["rgbs"] = b["rgbs"] * 2
    sd2 = sb.refine_slf(sd, scene, views, dev)
    np.testing.assert_array_equal(sd2["weight"]["count"].numpy(), f["refined_count"])
    np.testing.assert_allclose(sd2["weight"]["radiance"].numpy(), f["refined_radiance"], rtol=2e-5, atol=1e-6)
    for b in views:
        b["rgbs"] = b["rgbs"] / 2
    # extract_emitter_ldr.py:77-115
    em = sb.extract_emitters(scene, g["verts"], g["faces"], views, threshold=float(f["threshold"]), device=dev)
    np.testing.assert_array_equal(em["is_emitter"].numpy(), f["is_emitter"])
    np.testing.assert_array_equal(em["emitter_vertices"].numpy(), f["emitter_vertices"])
    np.testing.assert_allclose(em["emitter_area"].numpy(), f["emitter_area"], rtol=1e-6)
    np.testing.assert_allclose(em["emitter_normal"].numpy(), f["emitter_normal"], atol=1e-6)
    assert em["emitter_radiance"].shape == (len(g["faces"]), 3) and float(em["emitter_radiance"].abs().sum()) == 0.0
