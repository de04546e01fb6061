"""The stages BEFORE the bake, on the device: they write the two files the bake loads.

  bake_slf         slf_bake.py:69-145          scene bounds -> occupancy histogram -> VoxelSLF mean-pooled radiance -> vslf.npz
  refine_slf       slf_refine.py:85-108        re-pool an existing grid with new radiance
  extract_emitters extract_emitter_ldr.py:72-115  per-triangle mean radiance -> threshold -> emitter.pth

`views` is any iterable of dicts with 'rays' (N,6) [origin, direction] and, where radiance is pooled, 'rgbs' (N,3) LINEAR radiance:
the reference obtains it with ``model_crf.inverse(rgbs, exposure)`` (slf_bake.py:128-129) -- the learned camera response model and
the image datasets are outside this path, the caller applies them.  Primary rays are traced with the HIP intersector; pooling is
atomic accumulation in HBM (utils/gbuffer.py, VoxelSLF.scatter_add), so float sums agree with the reference's sequential CPU
scatter up to summation order, integer results exactly.  Everything stays on the device; only the final state dict goes to the
host, in the reference's file format (so SLFEmitter(emitter_path, slf_path) loads either side's files).
"""
import torch
import torch.nn.functional as NF

from . import _lib as L
from .model.slf import VoxelSLF
from .utils.gbuffer import scatter_add_rows, voxel_histogram
from .utils.path_tracing import ray_intersect


def _same_device(a, b):
    a, b = torch.device(a), torch.device(b)       # 'cuda' and 'cuda:0' are the same place: compare type and resolved index
    idx = lambda d: d.index if d.index is not None else (torch.cuda.current_device() if d.type == "cuda" else 0)
    return a.type == b.type and idx(a) == idx(b)


def _hits(scene, batch, device, cache=False):
    """Primary hits of one view: (positions, triangle index int32, valid).  The reference re-traces every view in each of its passes (bounds,
    histogram, pooling).  With cache=True the result is kept with the batch dict (28 B per pixel, i.e. 58 MB per 1080p view -- opt-in, because
    it pins that memory and the scene for as long as the caller keeps its batches; drop_hits() releases it) so a view is traced once."""
    hit = batch.get("_iris_hits") if (cache and isinstance(batch, dict)) else None
    if hit is None or not _same_device(hit[0].device, device) or hit[3] is not scene:
        rays = batch["rays"].to(device)
        positions, _, _, idx, valid = ray_intersect(scene, rays[..., :3].contiguous(), rays[..., 3:6].contiguous())
        hit = (positions, idx.to(torch.int32), valid, scene)
        if cache and isinstance(batch, dict):
            batch["_iris_hits"] = hit
    return hit[0], hit[1].long(), hit[2]


def drop_hits(views):
    """Release the primary hits _hits(cache=True) attached to the batches."""
    for batch in views:
        if isinstance(batch, dict):
            batch.pop("_iris_hits", None)


def scene_bounds(scene, views, dataset, device, cache=False):
    """slf_bake.py:69-93 including its scannetpp centre (``voxel_c = voxel_min + voxel_max``, not halved: kept, the files depend on it)."""
    # running min / max stay on the device (one host sync at the end instead of three per view); misses are masked out with the
    # identities of the reference's start values (1000, 0)
    lo = torch.tensor(1000.0, device=device)
    hi = torch.tensor(0.0, device=device)
    for batch in views:
        positions, _, valid = _hits(scene, batch, device, cache)
        v = valid[:, None]
        lo = torch.minimum(lo, torch.where(v, positions, lo).min())
        hi = torch.maximum(hi, torch.where(v, positions, hi).max())
    voxel_min, voxel_max = lo.cpu(), hi.cpu()
    if dataset in ("synthetic", "real"):
        voxel_min = 1.1 * voxel_min
        voxel_max = 1.1 * voxel_max
    else:
        voxel_c = voxel_min + voxel_max
        voxel_min, voxel_max = voxel_c + (voxel_min - voxel_c) * 1.1, voxel_c + (voxel_max - voxel_c) * 1.1
    return torch.as_tensor(voxel_min, dtype=torch.float32), torch.as_tensor(voxel_max, dtype=torch.float32)


def visible_voxels(scene, views, voxel_min, voxel_max, res_spatial, device, cache=False):
    """slf_bake.py:95-114: SpatialHist (res,res,res) f32 [z,y,x]; mask = SpatialHist > 0."""
    hist = torch.zeros(res_spatial, res_spatial, res_spatial, device=device, dtype=torch.float32)
    for batch in views:
        positions, _, valid = _hits(scene, batch, device, cache)
        voxel_histogram(positions[valid], float(voxel_min), float(voxel_max), res_spatial, hist)      # (an empty selection is a no-op)
    return hist


def pool_radiance(scene, views, vslf, device, cache=False):
    """slf_bake.py:120-138 / slf_refine.py:90-106: scatter every valid primary hit's radiance into its voxel, then average."""
    vslf = vslf.to(device)
    for batch in views:
        positions, _, valid = _hits(scene, batch, device, cache)
        vslf.scatter_add(positions[valid], batch["rgbs"].to(device=device, dtype=torch.float32)[valid])
    vslf.radiance = vslf.radiance / vslf.count[..., None].float().clamp_min(1)
    vslf.refresh()                                    # the buffer was rebound: the device tables are rebuilt at the next lookup
    return vslf


def bake_slf(scene, views, res_spatial=256, dataset="scannetpp", device="cuda"):
    """-> the dict slf_bake.py:140-145 saves as vslf.npz: {'mask', 'voxel_min', 'voxel_max', 'weight'}  (tensors on the host)."""
    views = list(views)
    device = torch.device(device)
    try:                                              # the three passes share one primary trace per view; nothing stays attached to the batches
        voxel_min, voxel_max = scene_bounds(scene, views, dataset, device, cache=True)
        hist = visible_voxels(scene, views, voxel_min, voxel_max, res_spatial, device, cache=True)
        mask = hist > 0
        vslf = VoxelSLF(mask, voxel_min.item(), voxel_max.item())            # index grid and buffers built on the device the mask was counted on
        vslf = pool_radiance(scene, views, vslf, device, cache=True)
    finally:
        drop_hits(views)
    return {"mask": mask.cpu(), "voxel_min": voxel_min.item(), "voxel_max": voxel_max.item(),
            "weight": {k: v.cpu() for k, v in vslf.state_dict().items()}}


def refine_slf(state_dict, scene, views, device="cuda"):
    """slf_refine.py:85-108: same grid, radiance pooled anew; returns the updated dict."""
    vslf = VoxelSLF(state_dict["mask"], state_dict["voxel_min"], state_dict["voxel_max"])
    vslf = pool_radiance(scene, list(views), vslf, torch.device(device))
    out = dict(state_dict)
    out["weight"] = {k: v.cpu() for k, v in vslf.state_dict().items()}
    return out


def extract_emitters(scene, vertices, faces, views, threshold, device="cuda"):
    """extract_emitter_ldr.py:77-115 (mode 'export'): per-triangle mean of the radiance of the primary hits that landed on it, max over
    the channels, > threshold -> emitter; returns the dict saved as emitter.pth."""
    device = torch.device(device)
    vertices = torch.as_tensor(vertices, dtype=torch.float32); faces = torch.as_tensor(faces).long()
    n_face = len(faces)
    triangle_radiance = torch.zeros(n_face, 3, device=device)
    triangle_count = torch.zeros(n_face, device=device)
    for batch in views:
        _, idx, valid = _hits(scene, batch, device)
        scatter_add_rows(batch["rgbs"].to(device=device, dtype=torch.float32)[valid], idx[valid], triangle_radiance, triangle_count)
    mean = triangle_radiance / triangle_count.unsqueeze(-1).clamp_min(1)
    is_emitter = (torch.max(mean, dim=-1)[0] > threshold).cpu()
    ev = vertices[faces[is_emitter]]
    area = torch.cross(ev[:, 1] - ev[:, 0], ev[:, 2] - ev[:, 0], dim=-1)
    return {"is_emitter": is_emitter, "emitter_vertices": ev, "emitter_area": area.norm(dim=-1) / 2.0, "emitter_normal": NF.normalize(area, dim=-1),
            "emitter_radiance": torch.zeros(n_face, 3)}
