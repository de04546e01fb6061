"""Shading helpers of the hot path that are callable on their own (reference: utils/ops.py)."""
import torch

from .. import _lib as L


def lerp_specular(specular, roughness):
    """Interpolate the 6 baked specular levels by roughness (utils/ops.py:99-118).
    specular: Bx6x3, roughness: Bx1 in [0.02,1.0] -> Bx3."""
    specular = L.require_gpu(specular, torch.float32, "specular")
    roughness = L.require_gpu(roughness, torch.float32, "roughness").reshape(-1)
    B, R, _ = specular.shape
    out = torch.empty(B, 3, device=specular.device, dtype=torch.float32)
    with torch.cuda.device(specular.device):
        L.check(L.lib().iris_lerp_specular(L.ptr(specular), L.ptr(roughness), B, R, L.ptr(out), L.stream()))
    return out
