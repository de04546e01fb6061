"""Pixel-centre camera rays, OpenCV convention (reference: utils/dataset/real_ldr.py:49-83 and the ScanNet++
loader utils/dataset/scannetpp/dataset.py:202-215).  The reference builds a meshgrid on the CPU, multiplies
by the rotation and copies 50 MB per 1080p view to the GPU; here 21 scalars go to one kernel."""
import ctypes as C

import numpy as np
import torch

from ... import _lib as L


class CameraDirections:
    """What get_direction returns: the description of the (H*W,3) camera-space directions, not the tensor."""

    def __init__(self, k, img_hw):
        self.k = L.host_f32(k).reshape(3, 3)
        self.img_hw = (int(img_hw[0]), int(img_hw[1]))


def get_direction(k, img_hw):
    """camera ray directions (unnormalised) for a 3x3 intrinsic matrix (real_ldr.py:49-61)"""
    return CameraDirections(k, img_hw)


def to_world(rays_d, c2w, ray_diff, k=None, device=None):
    """world-space origins and directions (real_ldr.py:63-83).
    rays_d: the CameraDirections from get_direction; c2w: 3x4.  Returns (rays_x, rays_d) normalised, or
    (rays_x, rays_d, dxdu, dydv) with un-normalised rays_d when ray_diff."""
    if not isinstance(rays_d, CameraDirections):
        raise L.IrisError("to_world: pass the object returned by get_direction")
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    H, W = rays_d.img_hw
    K = np.ascontiguousarray(rays_d.k if k is None else L.host_f32(k).reshape(3, 3), dtype=np.float32)
    M = L.host_f32(c2w).reshape(-1)[:12].copy()
    o = torch.empty(H * W, 3, device=device, dtype=torch.float32)
    d = torch.empty(H * W, 3, device=device, dtype=torch.float32)
    dx = torch.empty(H * W, 3, device=device, dtype=torch.float32) if ray_diff else None
    dy = torch.empty(H * W, 3, device=device, dtype=torch.float32) if ray_diff else None
    with torch.cuda.device(device):
        L.check(L.lib().iris_raygen_real(K.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p), H, W, int(bool(ray_diff)),
                                         L.ptr(o), L.ptr(d), L.ptr(dx), L.ptr(dy), L.stream()))
    return (o, d, dx, dy) if ray_diff else (o, d)
