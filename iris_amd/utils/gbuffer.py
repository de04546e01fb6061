"""Pooling primitives of the stages that produce the bake's inputs: the voxel occupancy
histogram of slf_bake.py:96-114 and the per-triangle radiance sums of extract_emitter_ldr.py:82-98.  (VoxelSLF.scatter_add is a
method of iris_amd.model.slf.VoxelSLF.)  Atomic accumulation on the GPU; float sums agree with the reference's sequential
scatter_add up to summation order."""
import torch

from .. import _lib as L


def voxel_histogram(position, voxel_min, voxel_max, res_spatial, hist=None):
    """SpatialHist of slf_bake.py:104-110: counts of positions per voxel, shape (H,H,H) indexed [z,y,x]; pass `hist` to
    accumulate over views."""
    position = L.require_gpu(position, torch.float32, "position").reshape(-1, 3)
    H = int(res_spatial)
    if hist is None:
        hist = torch.zeros(H, H, H, device=position.device, dtype=torch.float32)
    hist = L.require_gpu(hist, torch.float32, "hist")
    with torch.cuda.device(position.device):
        L.check(L.lib().iris_voxel_histogram(L.ptr(position), position.shape[0], float(voxel_min), float(voxel_max), H, L.ptr(hist), L.stream()))
    return hist


def scatter_add_rows(values, index, out, count=None):
    """out[index[i]] += values[i] (rows of 3), count[index[i]] += 1: torch_scatter.scatter(values, index, 0, out, reduce='sum')
    of extract_emitter_ldr.py:90-95."""
    values = L.require_gpu(values, torch.float32, "values").reshape(-1, 3)
    index = L.require_gpu(index, torch.int64, "index").reshape(-1)
    out = L.require_gpu(out, torch.float32, "out")
    if count is not None:
        count = L.require_gpu(count, torch.float32, "count")
    with torch.cuda.device(values.device):
        L.check(L.lib().iris_scatter_add_rows(L.ptr(values), L.ptr(index), values.shape[0], out.shape[0], L.ptr(out), L.ptr(count), L.stream()))
    return out, count
