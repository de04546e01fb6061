"""Deterministic stand-in for the reference's NGPBRDF (tiny-cuda-nn hash grid + MLP: third party):
same forward(position) -> {'albedo' Bx3, 'roughness' Bx1, 'metallic' Bx1} contract (model/brdf.py:243-260), closed form.
Evaluated with torch on the CPU everywhere (golden generation, oracle, GPU tests) so that all three see identical inputs."""
import numpy as np
import torch


class StubMaterial(torch.nn.Module):
    def forward(self, x):
        dev = x.device
        xc = x.detach().to("cpu", torch.float32)
        k = torch.tensor([1.3, 2.1, 0.7]); ph = torch.tensor([0.1, 0.5, 0.9])
        albedo = 0.5 + 0.4 * torch.sin(xc * k + ph)
        rough = 0.35 + 0.3 * torch.sin(xc[:, :1] * 1.7 + xc[:, 1:2] * 0.9)
        metal = 0.5 + 0.5 * torch.sin(xc[:, 2:3] * 2.3)
        return {"albedo": albedo.to(dev), "roughness": rough.to(dev), "metallic": metal.to(dev)}


def stub_material_np(position):
    out = StubMaterial()(torch.from_numpy(np.ascontiguousarray(position, dtype=np.float32)))
    return {k: v.numpy() for k, v in out.items()}


def material(voxel_min=None, voxel_max=None, ckpt=None):
    """factory with the signature `python -m iris_amd.refine_shading --material stub_material:material` expects (NGPBRDF(voxel_min, voxel_max))"""
    return StubMaterial()
