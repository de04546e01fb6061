"""SLFEmitter: triangle emitters + diffuse radiance cache (reference: model/emitter.py:134-221)."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as NF

from .. import _lib as L
from .slf import VoxelSLF


class SLFEmitter(nn.Module):
    """Loads the reference's ``emitter.pth`` / ``vslf.npz`` files (extract_emitter_ldr.py:109-115, slf_bake.py:140-145)."""

    def __init__(self, emitter_path, slf_path):
        super().__init__()
        state_dict = torch.load(slf_path, map_location="cpu")
        self.slf = VoxelSLF(state_dict["mask"], state_dict["voxel_min"], state_dict["voxel_max"])
        self.slf.load_state_dict(state_dict["weight"])

        weight = torch.load(emitter_path, map_location="cpu")
        is_emitter = weight["is_emitter"]
        self.register_buffer("is_emitter", is_emitter)
        self.register_buffer("emitter_vertices", weight["emitter_vertices"])
        self.register_buffer("emitter_area", weight["emitter_area"])
        self.register_buffer("radiance", weight["emitter_radiance"])
        emitter_idx = torch.full((len(is_emitter),), -1, dtype=torch.long)
        emitter_idx[is_emitter] = torch.arange(int(is_emitter.sum()))
        self.register_buffer("emitter_idx", emitter_idx)
        self.register_buffer("triangle_idx", torch.arange(len(is_emitter))[is_emitter])
        emitter_pdf = NF.normalize(torch.ones_like(weight["emitter_area"]), dim=-1, p=1)
        self.register_buffer("emitter_pdf", emitter_pdf)
        self.register_buffer("emitter_cdf", emitter_pdf.cumsum(-1).contiguous())
        self._h = None
        self._h_device = None
        self._struct_ver = None
        self._rad_version = None

    def refresh(self):
        h, self._h = self._h, None
        if h:
            L.lib().iris_emitter_destroy(h)

    @staticmethod
    def _ver(t):
        return (t._version, t.data_ptr(), str(t.device), tuple(t.shape))

    def handle(self, device):
        """Device-side tables of this emitter.  They follow the module: the tables are rebuilt when is_emitter / emitter_area /
        emitter_vertices change (load_state_dict, in-place edits, .to()) and the radiance table is re-uploaded when `radiance`
        does (SLFEmitterLearn's optimiser steps, model/emitter.py:268), so a cached handle never goes stale."""
        device = torch.device(device)
        struct = (self._ver(self.is_emitter), self._ver(self.emitter_area), self._ver(self.emitter_vertices))
        if self._h is None or self._h_device != device or self._struct_ver != struct:
            self.refresh()
            ie = np.ascontiguousarray(self.is_emitter.detach().cpu().numpy(), dtype=np.uint8)
            rad = L.host_f32(self.radiance).reshape(-1, 3)
            area = L.host_f32(self.emitter_area).reshape(-1)
            verts = L.host_f32(self.emitter_vertices).reshape(-1)
            cdf = L.host_f32(self.emitter_cdf).reshape(-1)
            has_v = verts.size == area.shape[0] * 9 and area.shape[0] > 0
            h = C.c_void_p()
            L.check(L.lib().iris_emitter_create(ie.ctypes.data_as(C.c_void_p), ie.shape[0], rad.ctypes.data_as(C.c_void_p), rad.shape[0],
                                                area.ctypes.data_as(C.c_void_p), area.shape[0],
                                                verts.ctypes.data_as(C.c_void_p) if has_v else None, cdf.ctypes.data_as(C.c_void_p) if has_v else None,
                                                L.device_index(device), C.byref(h)))
            self._h, self._h_device, self._struct_ver = h, device, struct
            self._rad_version = self._ver(self.radiance)
        elif self._rad_version != self._ver(self.radiance):
            rr = self.radiance_on(device)
            with torch.cuda.device(device):
                L.check(L.lib().iris_emitter_set_radiance(self._h, L.ptr(rr), rr.shape[0], L.stream()))
            self._rad_version = self._ver(self.radiance)
        return self._h

    def radiance_on(self, device):
        """`radiance` as a detached, contiguous float32 tensor on `device` (the files are loaded with map_location='cpu')."""
        return self.radiance.detach().to(device=device, dtype=torch.float32).contiguous()

    def __del__(self):
        try:
            self.refresh()
        except Exception:
            pass

    def forward(self, position):
        """surface light field from queried location (model/emitter.py:175-178)"""
        return self.slf(position)["rgb"]

    def eval_emitter(self, position, light_dir, triangle_idx, roughness=None, trace_roughness=0.6):
        """surface emission / radiance cache / path termination (model/emitter.py:180-221).
        Returns Le Bx3, emit_pdf Bx1, valid_next B (bool).  ``light_dir`` is unused, as in the reference."""
        position = L.require_gpu(position, torch.float32, "position").reshape(-1, 3)
        triangle_idx = L.require_gpu(triangle_idx, torch.int64, "triangle_idx").reshape(-1)
        B = position.shape[0]
        r = None
        if roughness is not None:  # bake passes an int64 tensor of ones (bake_shading.py:121)
            r = roughness.reshape(-1).to(device=position.device, dtype=torch.float32).contiguous()
        Le = torch.empty(B, 3, device=position.device, dtype=torch.float32)
        pdf = torch.empty(B, 1, device=position.device, dtype=torch.float32)
        vn = torch.empty(B, device=position.device, dtype=torch.bool)
        with torch.cuda.device(position.device):
            L.check(L.lib().iris_eval_emitter(self.handle(position.device), self.slf.handle(position.device), L.ptr(position),
                                              L.ptr(triangle_idx), L.ptr(r), float(trace_roughness), B, L.ptr(Le), L.ptr(pdf), L.ptr(vn), L.stream()))
        return Le, pdf, vn

    def sample_emitter(self, sample1, sample2, position):
        """importance sampling emitters (model/emitter.py:224-255): uniform emitter pick through the cdf, uniform point on
        the triangle.  Returns wi Bx3, pdf Bx1 (area measure), triangle_idx B."""
        sample1 = L.require_gpu(sample1, torch.float32, "sample1").reshape(-1)
        sample2 = L.require_gpu(sample2, torch.float32, "sample2").reshape(-1, 2)
        position = L.require_gpu(position, torch.float32, "position").reshape(-1, 3)
        B = position.shape[0]
        wi = torch.empty(B, 3, device=position.device, dtype=torch.float32)
        pdf = torch.empty(B, 1, device=position.device, dtype=torch.float32)
        tri = torch.empty(B, device=position.device, dtype=torch.int64)
        with torch.cuda.device(position.device):
            L.check(L.lib().iris_sample_emitter(self.handle(position.device), L.ptr(sample1), L.ptr(sample2), L.ptr(position), B, L.ptr(wi), L.ptr(pdf),
                                                L.ptr(tri), L.stream()))
        return wi, pdf, tri


class SLFEmitterLearn(SLFEmitter):
    """triangle emitters with a learnable radiance table (model/emitter.py:257-275)"""

    def __init__(self, emitter_path, slf_path):
        super().__init__(emitter_path, slf_path)
        rad = self.radiance
        del self._buffers["radiance"]
        self.radiance = nn.Parameter(torch.FloatTensor(rad))

    def update_slf(self, slf_path):
        state_dict = torch.load(slf_path, map_location="cpu")
        self.slf.load_state_dict(state_dict["weight"])
