"""The baked shading maps as ONE table resident in HBM, and the BRDF trainer's shading combine on it.

Reference:
  * utils/dataset/scannetpp/dataset.py:359-377 reads the 13 EXR maps of every view and concatenates them to
    ``all_cache (pixels, 39)`` = diffuse(3) | specular0 levels 0..5 (18) | specular1 levels 0..5 (18), on the host;
  * ``__getitem__`` (:409-414) slices a batch by a random pixel permutation into diffuse (B,3), specular0 (B,6,3), specular1 (B,6,3);
  * train_brdf_crf.py:195-203 combines them with the material net's outputs:
    ``kd = albedo*(1-metallic); ks = 0.04*(1-metallic)+albedo*metallic; L = kd*diffuse + ks*lerp(spec0,r) + lerp(spec1,r)``.
Here the table lives in HBM (1080p x 39 floats = 323 MB per view; 288 GB holds several hundred views), is filled directly from
the bake outputs (no EXR round trip) or from an EXR directory, and slice + combine + its gradient are one HIP kernel each
(iris_amd/csrc/iris_cache.h).  Row layout differs from the reference's (padded, level-interleaved); ``gather`` returns the
reference's slices.
"""
import os

import numpy as np
import torch

from .. import _lib as L
from . import exr


class ShadingCache:
    def __init__(self, n_pixels, roughness_level=6, device="cuda"):
        self.R = int(roughness_level)
        self.row_floats = int(L.lib().iris_cache_row_floats(self.R))
        if self.row_floats <= 0:
            raise ValueError("roughness_level must be in [1, 8]")
        self.rows = torch.empty(int(n_pixels), self.row_floats, device=device, dtype=torch.float32)

    def __len__(self):
        return self.rows.shape[0]

    def put_view(self, offset, diffuse, specular0, specular1):
        """Pack the maps of one view ((H*W,3) or (H,W,3) device tensors; specular0/1: lists of R maps, or (R,...,3) tensors) into
        rows [offset, offset + H*W)  (dataset.py:359-377)."""
        dev = self.rows.device
        d = L.require_gpu(diffuse.reshape(-1, 3), torch.float32, "diffuse")
        s0 = [L.require_gpu(m.reshape(-1, 3), torch.float32, "specular0") for m in specular0]
        s1 = [L.require_gpu(m.reshape(-1, 3), torch.float32, "specular1") for m in specular1]
        n = d.shape[0]
        if len(s0) != self.R or len(s1) != self.R or any(m.shape[0] != n for m in s0 + s1):
            raise ValueError("expected %d specular0 and specular1 maps of %d pixels" % (self.R, n))
        if offset < 0 or offset + n > len(self):
            raise ValueError("view does not fit the cache")
        import ctypes as C
        a0 = (C.c_void_p * self.R)(*[L.ptr(m) for m in s0])
        a1 = (C.c_void_p * self.R)(*[L.ptr(m) for m in s1])
        with torch.cuda.device(dev):
            L.check(L.lib().iris_cache_pack(L.ptr(d), a0, a1, n, self.R, self.rows[offset:].data_ptr(), L.stream()))
        return n

    @classmethod
    def from_bake(cls, views, device=None):
        """views: list of bake_shading.bake_view() results ({'diffuse', 'specular0': [R maps], 'specular1': [R maps]}): the maps go
        from the bake kernel's outputs straight into the table, no EXR round trip."""
        n = [v["diffuse"].reshape(-1, 3).shape[0] for v in views]
        c = cls(sum(n), len(views[0]["specular0"]), device if device is not None else views[0]["diffuse"].device)
        off = 0
        for v in views:
            off += c.put_view(off, v["diffuse"], v["specular0"], v["specular1"])
        return c

    @classmethod
    def from_exr_dir(cls, cache_dir, n_views, roughness_level=6, device="cuda"):
        """Read OUTPUT/diffuse/{:03d}.exr and OUTPUT/specular/{:03d}_{0,1}_{level}.exr (bake_shading.py:131,202-203)."""
        first = exr.read_exr(os.path.join(cache_dir, "diffuse", "%03d.exr" % 0))
        hw = first.shape[0] * first.shape[1]
        c = cls(hw * n_views, roughness_level, device)
        # the files of a few views are inflated concurrently (zlib and the large numpy operations release the GIL): a ZIP map takes ~0.2 s
        from concurrent.futures import ThreadPoolExecutor
        rd = lambda *p: np.ascontiguousarray(exr.read_exr(os.path.join(cache_dir, *p)))
        up = lambda a: torch.from_numpy(a).to(device)
        with ThreadPoolExecutor(max_workers=max(1, min(32, os.cpu_count() or 4))) as pool:
            def submit(i):
                return (pool.submit(rd, "diffuse", "%03d.exr" % i),
                        [pool.submit(rd, "specular", "%03d_0_%d.exr" % (i, j)) for j in range(roughness_level)],
                        [pool.submit(rd, "specular", "%03d_1_%d.exr" % (i, j)) for j in range(roughness_level)])
            ahead = 2                                                # views being read while view i is uploaded: bounds the host memory
            pending = [submit(i) for i in range(min(ahead, n_views))]
            for i in range(n_views):
                d, s0, s1 = pending.pop(0)
                if i + ahead < n_views:
                    pending.append(submit(i + ahead))
                c.put_view(i * hw, up(d.result()), [up(f.result()) for f in s0], [up(f.result()) for f in s1])
        return c

    def gather(self, idx=None):
        """The loader's batch slice (dataset.py:409-414): (diffuse (B,3), specular0 (B,R,3), specular1 (B,R,3))."""
        idx_t, B = self._idx(idx)
        out = torch.empty(B, 3 + 6 * self.R, device=self.rows.device, dtype=torch.float32)
        with torch.cuda.device(self.rows.device):
            L.check(L.lib().iris_cache_gather(L.ptr(self.rows), L.ptr(idx_t) if idx_t is not None else None, B, self.R, L.ptr(out), L.stream()))
        R3 = 3 * self.R
        return out[:, :3], out[:, 3:3 + R3].reshape(B, self.R, 3), out[:, 3 + R3:].reshape(B, self.R, 3)

    def _idx(self, idx):
        if idx is None:
            return None, len(self)
        idx_t = L.require_gpu(idx.reshape(-1), torch.int64, "idx")
        return idx_t, idx_t.shape[0]

    def shade(self, idx, albedo, metallic, roughness):
        """L (B,3) of train_brdf_crf.py:195-203 for the pixels idx (None = all rows in order); differentiable in albedo (B,3),
        metallic (B,1), roughness (B,1)."""
        idx_t, B = self._idx(idx)
        return _ShadeCached.apply(albedo, metallic, roughness, self.rows, idx_t, self.R)


class _ShadeCached(torch.autograd.Function):
    @staticmethod
    def forward(ctx, albedo, metallic, roughness, rows, idx, R):
        a = L.require_gpu(albedo.detach(), torch.float32, "albedo")
        m = L.require_gpu(metallic.detach().reshape(-1), torch.float32, "metallic")
        r = L.require_gpu(roughness.detach().reshape(-1), torch.float32, "roughness")
        B = a.shape[0]
        if m.shape[0] != B or r.shape[0] != B or (idx is not None and idx.shape[0] != B) or (idx is None and rows.shape[0] != B):
            raise ValueError("shade: albedo / metallic / roughness / idx disagree on the batch size")
        out = torch.empty(B, 3, device=rows.device, dtype=torch.float32)
        with torch.cuda.device(rows.device):
            L.check(L.lib().iris_shade_cached_fwd(L.ptr(rows), L.ptr(idx) if idx is not None else None, L.ptr(a), L.ptr(m), L.ptr(r), B, R,
                                                  L.ptr(out), L.stream()))
        ctx.save_for_backward(a, m, r, rows, idx if idx is not None else torch.empty(0, device=rows.device, dtype=torch.int64))
        ctx.has_idx = idx is not None
        ctx.R = R
        ctx.shapes = (metallic.shape, roughness.shape)
        return out

    @staticmethod
    def backward(ctx, gL):
        a, m, r, rows, idx = ctx.saved_tensors
        B = a.shape[0]
        gL = L.require_gpu(gL, torch.float32, "gL")
        need = ctx.needs_input_grad
        ga = torch.empty(B, 3, device=rows.device, dtype=torch.float32) if need[0] else None
        gm = torch.empty(B, device=rows.device, dtype=torch.float32) if need[1] else None
        gr = torch.empty(B, device=rows.device, dtype=torch.float32) if need[2] else None
        with torch.cuda.device(rows.device):
            L.check(L.lib().iris_shade_cached_bwd(L.ptr(rows), L.ptr(idx) if ctx.has_idx else None, L.ptr(a), L.ptr(m), L.ptr(r), L.ptr(gL), B,
                                                  ctx.R, L.ptr(ga) if need[0] else None, L.ptr(gm) if need[1] else None,
                                                  L.ptr(gr) if need[2] else None, L.stream()))
        ms, rs = ctx.shapes
        return ga, (gm.reshape(ms) if need[1] else None), (gr.reshape(rs) if need[2] else None), None, None, None
