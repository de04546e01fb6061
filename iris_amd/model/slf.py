"""VoxelSLF: sparse voxel radiance cache (reference: model/slf.py:16-70)."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L


class VoxelSLF(nn.Module):
    """voxel grid based surface light field; same constructor, buffers and state_dict keys as the reference."""

    def __init__(self, mask, voxel_min, voxel_max):
        super().__init__()
        H = mask.shape[0]
        self.H = H
        self.voxel_min = float(voxel_min)
        self.voxel_max = float(voxel_max)
        # (the buffers are built where the mask lives: a mask counted on the GPU -- slf_bake -- never visits the host; the reference builds on the CPU)
        mask = torch.as_tensor(mask)
        dev = mask.device
        kk, jj, ii = torch.where(mask)
        inds = -torch.ones(H, H, H, dtype=torch.long, device=dev)
        inds[kk, jj, ii] = torch.arange(len(ii), device=dev)
        self.register_buffer("inds", inds)
        self.register_buffer("radiance", torch.zeros(len(ii), 3, device=dev))
        self.register_buffer("count", torch.zeros(len(ii), dtype=torch.long, device=dev))
        self._h = None
        self._h_device = None
        self._ver = None
        self._rver = None

    # -- device handle ---------------------------------------------------------------------------------
    def refresh(self):
        """Drop the device-side tables; they are rebuilt from the buffers at the next lookup."""
        h, self._h = self._h, None
        if h:
            L.lib().iris_slf_destroy(h)

    @staticmethod
    def _tver(t):
        # identity + version of the uploaded tensor.  The tensor object itself is kept (not its data_ptr): a rebound buffer
        # (`vslf.radiance = vslf.radiance / n`) is a new object even when the caching allocator hands it the address -- and _version 0 -- of
        # the tensor that was uploaded before, and holding the reference keeps that address from being reused while the key is alive.
        return (t, t._version)

    @staticmethod
    def _same(a, b):
        return a is not None and b is not None and a[0] is b[0] and a[1] == b[1]

    def handle(self, device, need_radiance=True):
        """Device-side tables (int32 index grid + padded radiance rows).  They follow the module: the tables are rebuilt when `inds`
        changed since the upload (load_state_dict through a parent module, in-place edits, .to()), and the radiance rows alone are
        re-uploaded when only `radiance` did (scatter_add, mean pooling) -- lazily, the next time a lookup needs them."""
        device = torch.device(device)
        iv = self._tver(self.inds)
        if self._h is None or self._h_device != device or not self._same(self._ver, iv):
            self.refresh()
            h = C.c_void_p()
            # the device index resolved ONCE ('cuda' without an index = torch's current device): the same index decides whether the tensors are already
            # there, which device torch makes current around the call, and which device the C side creates the tables on
            idx = device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else 0)
            on_dev = lambda t: t.is_cuda and t.device.index == idx
            if device.type == "cuda" and on_dev(self.inds) and on_dev(self.radiance):
                inds = self.inds.detach().to(torch.int64).contiguous()
                rad = self.radiance.detach().to(torch.float32).contiguous().reshape(-1, 3)
                with torch.cuda.device(idx):
                    L.check(L.lib().iris_slf_create_dev(L.ptr(inds), self.H, L.ptr(rad), rad.shape[0], self.voxel_min, self.voxel_max, idx, C.byref(h), L.stream()))
            else:
                inds = np.ascontiguousarray(self.inds.detach().cpu().numpy(), dtype=np.int64)
                rad = L.host_f32(self.radiance).reshape(-1, 3)
                L.check(L.lib().iris_slf_create(inds.ctypes.data_as(C.c_void_p), self.H, rad.ctypes.data_as(C.c_void_p), rad.shape[0],
                                                self.voxel_min, self.voxel_max, idx, C.byref(h)))
            self._h, self._h_device, self._ver, self._rver = h, device, iv, self._tver(self.radiance)
        elif need_radiance and not self._same(self._rver, self._tver(self.radiance)):
            rr = self.radiance.detach().to(device=device, dtype=torch.float32).contiguous()
            with torch.cuda.device(device):
                L.check(L.lib().iris_slf_set_radiance(self._h, L.ptr(rr), rr.shape[0], L.stream()))
            self._rver = self._tver(self.radiance)
        return self._h

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.refresh()
        return r

    def __del__(self):
        try:
            self.refresh()
        except Exception:
            pass

    # -- reference API ---------------------------------------------------------------------------------
    def _lookup(self, x, want_idx, want_rgb):
        x = L.require_gpu(x, torch.float32, "x").reshape(-1, 3)
        B = x.shape[0]
        idx = torch.empty(B, device=x.device, dtype=torch.int64) if want_idx else None
        rgb = torch.empty(B, 3, device=x.device, dtype=torch.float32) if want_rgb else None
        with torch.cuda.device(x.device):
            L.check(L.lib().iris_slf_lookup(self.handle(x.device), L.ptr(x), B, L.ptr(idx), L.ptr(rgb), L.stream()))
        return idx, rgb

    def spatial_idx(self, x):
        """voxel entry index for Bx3 positions, -1 = empty (model/slf.py:41-54)"""
        return self._lookup(x, True, False)[0]

    def forward(self, x):
        """query surface light field; zero radiance in empty space (model/slf.py:63-70)"""
        return {"rgb": self._lookup(x, False, True)[1]}

    def scatter_add(self, x, radiance):
        """scatter add radiance into the voxel grid and count the entries (model/slf.py:56-61); the caller divides by
        count afterwards for mean pooling (slf_bake.py:138).  x, radiance: Bx3 on the GPU; the module's `radiance` / `count`
        buffers must live on the same device."""
        x = L.require_gpu(x, torch.float32, "x").reshape(-1, 3)
        radiance = L.require_gpu(radiance, torch.float32, "radiance").reshape(-1, 3)
        acc = L.require_gpu(self.radiance, torch.float32, "VoxelSLF.radiance buffer")
        cnt = L.require_gpu(self.count, torch.int64, "VoxelSLF.count buffer")
        with torch.cuda.device(x.device):
            L.check(L.lib().iris_slf_scatter_add(self.handle(x.device, need_radiance=False), L.ptr(x), L.ptr(radiance), x.shape[0], L.ptr(acc), L.ptr(cnt), L.stream()))
        # the kernel wrote the buffers behind torch's back: bump their version counters so that handle() re-uploads them
        self.radiance.add_(0); self.count.add_(0)
