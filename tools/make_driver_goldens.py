#!/usr/bin/env python3
"""Golden fixtures for the DRIVER LOOPS around the path (verdict round 2, item 6), produced by replaying the loop bodies of the reference's own
scripts through its imported Python (stub modules of tools/make_goldens.py; the oracle's closest hit stands in for Mitsuba):

  tests/golden/prebake_box.npz   slf_bake.py:69-138 (scene bounds incl. the non-halved scannetpp centre, occupancy histogram, VoxelSLF mean
                                 pooling through the reference's VoxelSLF class), slf_refine.py:90-106 (the same grid pooled anew), and
                                 extract_emitter_ldr.py:77-115 (per-triangle mean radiance -> emitter.pth).  Two things are stood in for and said so:
                                 torch_scatter.scatter(reduce='sum') (absent here) by Tensor.index_add_, its definition; the CRF inverse
                                 (crf/model_crf.py, out of scope) by the identity -- the views carry linear radiance.
  tests/golden/refine_loop.npz   refine_shading.py:109-127 (diffuse) and :144-174 (six specular levels) on one small view: the reference's
                                 path_tracing_det_diff / path_tracing_det_spec called exactly as the loops call them (its own batch_size), every
                                 torch.rand draw recorded per call; without the OptiX denoiser and cv2.imwrite.

    python tools/make_driver_goldens.py
"""
import math
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, os.path.join(REPO, "tests"))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def prebake_views(c2w0, H=48, W=64, n=3):
    """The four 'photographs' of the box room the pre-bake fixture uses: rays (N,6) per view (numpy; the same code builds them in the tests)."""
    import synth
    import oracle
    views = []
    for v in range(n):
        K, c2w = synth.camera(H, W, v, n_views=n)
        xs, ds = oracle.raygen_real(K, c2w if v else c2w0, H, W)
        views.append(np.concatenate([xs, ds], -1).astype(np.float32))
    gx, gy = np.meshgrid(np.linspace(0.2, 3.8, W, dtype=np.float32), np.linspace(0.2, 2.8, H, dtype=np.float32), indexing="xy")
    tgt = np.stack([gx, gy, np.full_like(gx, 2.6)], -1).reshape(-1, 3)
    o = np.broadcast_to(np.array([2.0, 1.5, 1.0], np.float32), tgt.shape)
    d = tgt - o
    d = (d / np.linalg.norm(d.astype(np.float64), axis=-1, keepdims=True)).astype(np.float32)
    views.append(np.concatenate([o, d], -1).astype(np.float32))
    return views


def main():
    import torch
    import torch.nn.functional as NF
    from make_goldens import _stub_modules
    _stub_modules()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from model.brdf import BaseBRDF
    from model.slf import VoxelSLF
    from model.emitter import SLFEmitter
    import utils.path_tracing as rpt
    from utils.dataset import real_ldr
    import oracle
    from stub_material import StubMaterial

    g = np.load(os.path.join(OUT, "bake_box.npz"))
    verts, faces = g["verts"], g["faces"]
    osc = oracle.Scene(verts, faces)

    def ray_intersect(scene, xs, ds):
        p, n, uv, idx, valid = osc.ray_intersect(xs.numpy(), ds.numpy())
        return (torch.from_numpy(p), torch.from_numpy(n), torch.from_numpy(uv), torch.from_numpy(idx), torch.from_numpy(valid))
    rpt.ray_intersect = ray_intersect

    # ================================================================== prebake_box.npz
    device = "cpu"
    rays_np = prebake_views(g["c2w"])
    dataset = []
    for r in rays_np:
        rays = torch.from_numpy(r)
        pos, _, _, idx, valid = ray_intersect(None, rays[..., :3], rays[..., 3:6])
        rgb = 0.3 + 0.2 * torch.sin(pos * 2.0)                 # linear radiance "photographed" in the view: smooth field, bright light quad
        rgb[idx >= 12] = torch.tensor([10.0, 9.0, 8.0])
        rgb[~valid] = 0
        dataset.append({"rays": rays, "rgbs": rgb})
    res_spatial = 24
    out = {"res_spatial": res_spatial, "n_views": len(dataset)}
    for k, b in enumerate(dataset):
        out[f"rays_{k}"] = b["rays"].numpy(); out[f"rgbs_{k}"] = b["rgbs"].numpy()

    # ---- slf_bake.py:69-93 scene bounds
    voxel_min = 1000.
    voxel_max = 0.0
    for idx in range(len(dataset)):
        batch = dataset[idx]
        rays = batch['rays']
        xs = rays[..., :3]
        ds = rays[..., 3:6]
        positions, _, _, _, valid = ray_intersect(None, xs.to(device), ds.to(device))
        if not valid.any():
            continue
        position = positions[valid]
        voxel_min = min(voxel_min, position.min())
        voxel_max = max(voxel_max, position.max())
    out["bounds_raw"] = np.array([float(voxel_min), float(voxel_max)], np.float32)
    out["bounds_synthetic"] = np.array([float(1.1 * voxel_min), float(1.1 * voxel_max)], np.float32)      # :86-88
    voxel_c = (voxel_min + voxel_max)                                                                     # :90 (scannetpp; not halved)
    voxel_min = voxel_c + (voxel_min - voxel_c) * 1.1
    voxel_max = voxel_c + (voxel_max - voxel_c) * 1.1
    # ---- slf_bake.py:95-114 visible voxels
    SpatialHist = torch.zeros(res_spatial ** 3, device=device)
    for idx in range(len(dataset)):
        batch = dataset[idx]
        rays = batch['rays']
        positions, _, _, _, valid = ray_intersect(None, rays[..., :3], rays[..., 3:6])
        if not valid.any():
            continue
        position = (positions[valid] - voxel_min) / (voxel_max - voxel_min)
        position = (position * res_spatial).long().clamp(0, res_spatial - 1)
        inds = position[..., 0] + position[..., 1] * res_spatial + position[..., 2] * res_spatial * res_spatial
        SpatialHist.scatter_add_(0, inds, torch.ones_like(inds).float())
    SpatialHist = SpatialHist.reshape(res_spatial, res_spatial, res_spatial)
    mask = (SpatialHist > 0)
    # ---- slf_bake.py:116-138 pooling (model_crf.inverse = identity here)
    vslf = VoxelSLF(mask.cpu(), voxel_min.item(), voxel_max.item())
    for idx in range(len(dataset)):
        batch = dataset[idx]
        rays = batch['rays']
        radiance = batch['rgbs']
        positions, _, _, _, valid = ray_intersect(None, rays[..., :3], rays[..., 3:6])
        if not valid.any():
            continue
        vslf.scatter_add(positions[valid].cpu(), radiance.to(device)[valid].cpu())
    vslf.radiance = vslf.radiance / vslf.count[..., None].float().clamp_min(1)
    out.update({"voxel_min": np.float64(voxel_min.item()), "voxel_max": np.float64(voxel_max.item()), "hist": SpatialHist.numpy(), "mask": mask.numpy(),
                "slf_inds": vslf.inds.numpy(), "slf_radiance": vslf.radiance.numpy(), "slf_count": vslf.count.numpy()})
    # ---- slf_refine.py:90-106: the same grid, radiance pooled anew (here: the photographs doubled)
    vslf2 = VoxelSLF(mask.cpu(), voxel_min.item(), voxel_max.item())
    for idx in range(len(dataset)):
        batch = dataset[idx]
        rays = batch['rays']
        radiance = batch['rgbs'] * 2
        positions, _, _, _, valid = ray_intersect(None, rays[..., :3], rays[..., 3:6])
        if not valid.any():
            continue
        vslf2.scatter_add(positions[valid].cpu(), radiance.to(device)[valid].cpu())
    vslf2.radiance = vslf2.radiance / vslf2.count[..., None].float().clamp_min(1)
    out.update({"refined_radiance": vslf2.radiance.numpy(), "refined_count": vslf2.count.numpy()})
    # ---- extract_emitter_ldr.py:77-115 (mode 'export'); torch_scatter.scatter(src, index, 0, out, reduce='sum') == out.index_add_(0, index, src)
    vertices = torch.from_numpy(verts).float()
    faces_t = torch.from_numpy(faces.astype(np.int64))
    n_face = len(faces_t)
    triangle_radiance = torch.zeros(n_face, 3)
    triangle_count = torch.zeros(n_face)
    for batch in dataset:
        rays = batch['rays']
        rays_x, rays_d = rays[..., :3].to(device), rays[..., 3:6].to(device)
        positions, normals, uvs, triangle_idxs, valid = ray_intersect(None, rays_x, rays_d)
        triangle_idxs = triangle_idxs[valid].cpu()
        radiance = batch['rgbs'][valid.cpu()]
        triangle_radiance = triangle_radiance.index_add_(0, triangle_idxs, radiance)
        triangle_count = triangle_count.index_add_(0, triangle_idxs, torch.ones(len(triangle_idxs)))
    triangle_radiance_mean = triangle_radiance / triangle_count.unsqueeze(-1).clamp_min(1)
    triangle_radiance_mean = torch.max(triangle_radiance_mean, dim=-1)[0]
    threshold = 5.0
    is_emitter = triangle_radiance_mean > threshold
    emitter_vertices = vertices[faces_t[is_emitter]]
    emitter_area = torch.cross(emitter_vertices[:, 1] - emitter_vertices[:, 0], emitter_vertices[:, 2] - emitter_vertices[:, 0], -1)
    emitter_normal = NF.normalize(emitter_area, dim=-1)
    emitter_area = emitter_area.norm(dim=-1) / 2.0
    out.update({"threshold": threshold, "triangle_radiance": triangle_radiance.numpy(), "triangle_count": triangle_count.numpy(), "is_emitter": is_emitter.numpy(),
                "emitter_vertices": emitter_vertices.numpy(), "emitter_area": emitter_area.numpy(), "emitter_normal": emitter_normal.numpy()})
    np.savez_compressed(os.path.join(OUT, "prebake_box.npz"), **out)
    print("prebake_box: bounds", out["bounds_raw"], "->", float(out["voxel_min"]), float(out["voxel_max"]), "occupied voxels", int(mask.sum()), "emitters", int(is_emitter.sum()))

    # ================================================================== refine_loop.npz
    p = np.load(os.path.join(OUT, "pt_single.npz"))
    tmp = tempfile.mkdtemp()
    emitter_path = os.path.join(tmp, "emitter.pth"); slf_path = os.path.join(tmp, "vslf.npz")
    bslf = VoxelSLF(torch.from_numpy(g["slf_mask"]), float(g["voxel_min"]), float(g["voxel_max"]))
    bslf.radiance[:] = torch.from_numpy(g["slf_radiance"])
    torch.save({"is_emitter": torch.from_numpy(g["is_emitter"]), "emitter_vertices": torch.from_numpy(p["emitter_vertices"]), "emitter_area": torch.from_numpy(g["emitter_area"]),
                "emitter_normal": torch.zeros(2, 3), "emitter_radiance": torch.from_numpy(p["radiance"])}, emitter_path)
    torch.save({"mask": torch.from_numpy(g["slf_mask"]), "voxel_min": float(g["voxel_min"]), "voxel_max": float(g["voxel_max"]), "weight": bslf.state_dict()}, slf_path)
    emitter = SLFEmitter(emitter_path, slf_path)                                  # refine_shading.py:75

    class RefStub(BaseBRDF):            # stands in for NGPBRDF (tiny-cuda-nn), same forward(position) contract (model/brdf.py:243-260)
        def forward(self, x):
            return StubMaterial()(x)
    material_net = RefStub()
    Hr, Wr = 10, 14
    Kr = torch.tensor([[0.8 * Wr, 0, Wr / 2], [0, 0.8 * Wr, Hr / 2], [0, 0, 1]], dtype=torch.float32)
    c2w = torch.from_numpy(g["c2w"])
    rays_x, rays_d = real_ldr.to_world(real_ldr.get_direction(Kr, (Hr, Wr)), c2w, False, Kr)
    real_rand = torch.rand

    def record(fn):
        rec = []

        def rr(*a, **k):
            k.pop("device", None)
            t = real_rand(*a, **k); rec.append(t.clone()); return t
        torch.rand = rr
        try:
            return fn(), rec
        finally:
            torch.rand = real_rand
    # (seeds 11 and 12 each draw one grazing sample whose origin + RayEpsilon * wi lands within an ulp of a wall plane: hit or miss -- and with it the
    #  number of continuing paths the next draws are sized for -- then depends on the last bit of wi, which torch and the oracle do not share; see pt_full.npz)
    seed = int(os.environ.get("IRIS_GOLDEN_REFINE_LOOP_SEED", "13"))
    torch.manual_seed(seed)
    ref = {"H": Hr, "W": Wr, "K": Kr.numpy(), "c2w": c2w.numpy(), "rays_x": rays_x.numpy(), "rays_d": rays_d.numpy(), "seed": seed}
    with torch.no_grad():
        # ---- refine_shading.py:99-127 (diffuse): spp and depth reduced, the loop itself as written
        spp = 8
        indir_depth = 2
        batch_size = 10240 * 128 // spp
        positions, normals, uvs, triangle_idxs, valid = ray_intersect(None, rays_x, rays_d)
        wi = rays_d
        B = len(positions)
        L = torch.zeros(B, 3)
        calls = []
        for b in range(math.ceil(B * 1.0 / batch_size)):
            b0 = b * batch_size
            b1 = min(b0 + batch_size, B)
            res, rec = record(lambda: rpt.path_tracing_det_diff(None, emitter, material_net, positions[b0:b1], wi[b0:b1], normals[b0:b1], uvs[b0:b1], triangle_idxs[b0:b1], spp, indir_depth))
            L[b0:b1] = res
            calls.append(rec)
        assert L.isnan().any() == False
        ref.update({"spp_diffuse": spp, "indir_depth": indir_depth, "diffuse": L.numpy(), "n_calls_0": len(calls), "valid": valid.numpy()})
        for c, rec in enumerate(calls):
            ref[f"n_u_0_{c}"] = len(rec)
            for k, t in enumerate(rec):
                ref[f"u_0_{c}_{k}"] = t.numpy()
        # ---- refine_shading.py:133-174 (specular)
        spp = 4
        batch_size = 10240 * 128 // spp
        roughness_level = torch.linspace(0.02, 1.0, 6)
        ref["spp_specular"] = spp
        for r_idx, roughness in enumerate(roughness_level):
            B = len(positions)
            L0 = torch.zeros(B, 3)
            L1 = L0.clone()
            calls = []
            for b in range(math.ceil(B * 1.0 / batch_size)):
                b0 = b * batch_size
                b1 = min(b0 + batch_size, B)
                (L0_, L1_), rec = record(lambda: rpt.path_tracing_det_spec(None, emitter, material_net, roughness, positions[b0:b1], wi[b0:b1], normals[b0:b1], uvs[b0:b1], triangle_idxs[b0:b1], spp, indir_depth))
                L0[b0:b1] = L0_
                L1[b0:b1] = L1_
                calls.append(rec)
            assert L0.isnan().any() == False
            assert L1.isnan().any() == False
            ref[f"specular0_{r_idx}"] = L0.numpy(); ref[f"specular1_{r_idx}"] = L1.numpy(); ref[f"n_calls_{r_idx + 1}"] = len(calls)
            for c, rec in enumerate(calls):
                ref[f"n_u_{r_idx + 1}_{c}"] = len(rec)
                for k, t in enumerate(rec):
                    ref[f"u_{r_idx + 1}_{c}_{k}"] = t.numpy()
    np.savez_compressed(os.path.join(OUT, "refine_loop.npz"), **ref)
    print("refine_loop: diffuse mean", float(ref["diffuse"].mean()), "draws per call", [ref[f"n_u_{l}_0"] for l in range(7)],
          os.path.getsize(os.path.join(OUT, "refine_loop.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
