"""BaseBRDF samplers (reference: model/brdf.py:20-136) as single fused gfx950 kernels."""
import torch
import torch.nn as nn

from .. import _lib as L


class BaseBRDF(nn.Module):
    """Parameter-free BRDF used by bake_shading (bake_shading.py:79)."""

    def __init__(self):
        super().__init__()

    def forward(self):
        pass

    def sample_diffuse(self, sample2, normal):
        """Cosine-weighted direction, pdf = relu(n.wi)/pi, weight = 1 (model/brdf.py:78-88)."""
        sample2 = L.require_gpu(sample2, torch.float32, "sample2").reshape(-1, 2)
        normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
        B = sample2.shape[0]
        wi = torch.empty(B, 3, device=sample2.device, dtype=torch.float32)
        pdf = torch.empty(B, 1, device=sample2.device, dtype=torch.float32)
        w = torch.empty(B, 3, device=sample2.device, dtype=torch.float32)
        with torch.cuda.device(sample2.device):
            L.check(L.lib().iris_sample_diffuse(L.ptr(sample2), L.ptr(normal), B, L.ptr(wi), L.ptr(pdf), L.ptr(w), L.stream()))
        return wi, pdf, w

    def sample_specular(self, sample2, wo, normal, roughness):
        """GGX half-vector sampling and the two Fresnel-split weights (model/brdf.py:112-136).
        roughness: python float or 0-d tensor (bake_shading.py:161 iterates a linspace)."""
        sample2 = L.require_gpu(sample2, torch.float32, "sample2").reshape(-1, 2)
        wo = L.require_gpu(wo, torch.float32, "wo").reshape(-1, 3)
        normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
        if isinstance(roughness, torch.Tensor):
            if roughness.numel() != 1:
                raise L.IrisError("sample_specular: per-sample roughness is not on the bake path; pass a scalar")
            roughness = float(roughness.detach().float().cpu().item())
        B = sample2.shape[0]
        wi = torch.empty(B, 3, device=sample2.device, dtype=torch.float32)
        pdf = torch.empty(B, 1, device=sample2.device, dtype=torch.float32)
        w0 = torch.empty(B, 1, device=sample2.device, dtype=torch.float32)
        w1 = torch.empty(B, 1, device=sample2.device, dtype=torch.float32)
        with torch.cuda.device(sample2.device):
            L.check(L.lib().iris_sample_specular(L.ptr(sample2), L.ptr(wo), L.ptr(normal), roughness, B, L.ptr(wi), L.ptr(pdf), L.ptr(w0), L.ptr(w1), L.stream()))
        return wi, pdf, w0, w1

    @staticmethod
    def _mat(mat, dev):
        albedo = L.require_gpu(mat["albedo"].detach(), torch.float32, "mat['albedo']").reshape(-1, 3)
        rough = L.require_gpu(mat["roughness"].detach(), torch.float32, "mat['roughness']").reshape(-1)
        metal = L.require_gpu(mat["metallic"].detach(), torch.float32, "mat['metallic']").reshape(-1)
        return albedo, rough, metal

    def eval_brdf(self, wi, wo, normal, mat):
        """BRDF value (already multiplied by NoL) and the 50/50 diffuse + GGX sampling pdf (model/brdf.py:138-175).
        mat: {'albedo' Bx3, 'roughness' Bx1, 'metallic' Bx1}.  Returns brdf Bx3, pdf Bx1."""
        wi = L.require_gpu(wi, torch.float32, "wi").reshape(-1, 3)
        wo = L.require_gpu(wo, torch.float32, "wo").reshape(-1, 3)
        normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
        albedo, rough, metal = self._mat(mat, wi.device)
        B = wi.shape[0]
        brdf = torch.empty(B, 3, device=wi.device, dtype=torch.float32)
        pdf = torch.empty(B, 1, device=wi.device, dtype=torch.float32)
        with torch.cuda.device(wi.device):
            L.check(L.lib().iris_eval_brdf(L.ptr(wi), L.ptr(wo), L.ptr(normal), L.ptr(albedo), L.ptr(rough), L.ptr(metal), B, L.ptr(brdf), L.ptr(pdf), L.stream()))
        return brdf, pdf

    def sample_brdf(self, sample1, sample2, wo, normal, mat):
        """importance sampling: diffuse lobe where sample1 > 0.5, GGX otherwise; returns wi Bx3, pdf Bx1, brdf/pdf Bx3
        (model/brdf.py:177-210)."""
        sample1 = L.require_gpu(sample1, torch.float32, "sample1").reshape(-1)
        sample2 = L.require_gpu(sample2, torch.float32, "sample2").reshape(-1, 2)
        wo = L.require_gpu(wo, torch.float32, "wo").reshape(-1, 3)
        normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
        albedo, rough, metal = self._mat(mat, wo.device)
        B = wo.shape[0]
        wi = torch.empty(B, 3, device=wo.device, dtype=torch.float32)
        pdf = torch.empty(B, 1, device=wo.device, dtype=torch.float32)
        w = torch.empty(B, 3, device=wo.device, dtype=torch.float32)
        with torch.cuda.device(wo.device):
            L.check(L.lib().iris_sample_brdf(L.ptr(sample1), L.ptr(sample2), L.ptr(wo), L.ptr(normal), L.ptr(albedo), L.ptr(rough), L.ptr(metal), B,
                                             L.ptr(wi), L.ptr(pdf), L.ptr(w), L.stream()))
        return wi, pdf, w
