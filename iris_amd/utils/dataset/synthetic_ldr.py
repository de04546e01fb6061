"""Pixel-centre camera rays of the synthetic (FIPT) datasets (reference: utils/dataset/synthetic_ldr.py:21-57)."""
import ctypes as C

import torch

from ... import _lib as L


class RayDirections:
    def __init__(self, H, W, focal):
        self.H, self.W, self.focal = int(H), int(W), float(focal)


def get_ray_directions(H, W, focal):
    """camera ray directions, x: left, y: up, z: forward (synthetic_ldr.py:21-34)"""
    return RayDirections(H, W, focal)


def get_rays(directions, c2w, focal=None, device=None):
    """world space camera rays (synthetic_ldr.py:36-57); focal not None -> also ray differentials."""
    if not isinstance(directions, RayDirections):
        raise L.IrisError("get_rays: pass the object returned by get_ray_directions")
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    H, W = directions.H, directions.W
    ray_diff = focal is not None
    M = L.host_f32(c2w).reshape(-1)[:12].copy()
    o = torch.empty(H * W, 3, device=device, dtype=torch.float32)
    d = torch.empty(H * W, 3, device=device, dtype=torch.float32)
    dx = torch.empty(H * W, 3, device=device, dtype=torch.float32) if ray_diff else None
    dy = torch.empty(H * W, 3, device=device, dtype=torch.float32) if ray_diff else None
    with torch.cuda.device(device):
        L.check(L.lib().iris_raygen_synthetic(float(focal) if ray_diff else directions.focal, M.ctypes.data_as(C.c_void_p), H, W,
                                              int(ray_diff), L.ptr(o), L.ptr(d), L.ptr(dx), L.ptr(dy), L.stream()))
    return (o, d, dx, dy) if ray_diff else (o, d)
