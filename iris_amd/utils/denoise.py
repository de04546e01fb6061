"""Denoiser that stands where ``mitsuba.OptixDenoiser`` stands in the bake (bake_shading.py:81,129,198-200).

The reference constructs ``OptixDenoiser(img_hw[::-1])`` once and calls it on every (H,W,3) map except the lowest roughness level.
The OptiX AI denoiser is closed and NVIDIA-only; this is a variance-guided edge-avoiding a-trous filter in HIP
(iris_amd/csrc/iris_denoise.h) guided by the primary-hit normal / position of the view, which the bake has anyway.  It cannot be
bit-compared with OptiX: tests check the HIP kernels against the oracle's restatement of the same filter and the PSNR gain against a
high-spp bake of the same view.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L


class Denoiser:
    def __init__(self, size_wh, device="cuda", iterations=5, sigma_l=16.0, sigma_n=128.0, sigma_p=0.05):
        self.W, self.H = int(size_wh[0]), int(size_wh[1])            # OptixDenoiser takes (width, height): bake_shading.py:81
        self.device = torch.device(device)
        self.iterations, self.sigma_l, self.sigma_n, self.sigma_p = int(iterations), float(sigma_l), float(sigma_n), float(sigma_p)
        self._ws = None
        self._normal = self._position = self._valid = None

    def set_guides(self, normal=None, position=None, valid=None):
        """Primary hits of the view in image order: normal / position (H*W,3) f32, valid (H*W,) bool (ray_intersect's outputs)."""
        n = self.H * self.W
        chk = lambda t, dt, name: None if t is None else L.require_gpu(t.reshape(n, -1) if dt != torch.uint8 else t.reshape(n), dt, name)
        self._normal = chk(normal, torch.float32, "normal")
        self._position = chk(position, torch.float32, "position")
        self._valid = None if valid is None else chk(valid.to(torch.uint8), torch.uint8, "valid")
        return self

    def denoise_maps(self, maps):
        """maps: list of (H,W,3) / (H*W,3) f32 device tensors -> list of new (H,W,3) tensors (filtered 4 at a time, sharing the guides)."""
        n = self.H * self.W
        ins = [L.require_gpu(m.reshape(n, 3), torch.float32, "map") for m in maps]
        outs = [torch.empty(n, 3, device=self.device, dtype=torch.float32) for _ in ins]
        if not ins:
            return []
        need = int(L.lib().iris_denoise_workspace_bytes(self.H, self.W))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        a_in = (C.c_void_p * len(ins))(*[m.data_ptr() for m in ins])
        a_out = (C.c_void_p * len(ins))(*[m.data_ptr() for m in outs])
        with torch.cuda.device(self.device):
            L.check(L.lib().iris_denoise(L.ptr(self._normal), L.ptr(self._position), L.ptr(self._valid), self.H, self.W, len(ins), a_in, a_out,
                                         self.iterations, self.sigma_l, self.sigma_n, self.sigma_p, L.ptr(self._ws), need, L.stream()))
        return [o.reshape(self.H, self.W, 3) for o in outs]

    def __call__(self, img):
        """One (H,W,3) image (device tensor, or a numpy array as the reference passes: uploaded) -> (H,W,3) device tensor."""
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).to(self.device)
        return self.denoise_maps([img])[0]
