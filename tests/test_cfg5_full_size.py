"""BASELINE.json configs[4] at its stated size on the GPU: train_emitter's inner loop (train_emitter.py:181-189) -- 4 calls of
path_tracing_single (utils/path_tracing.py:320-407) on 8 192 pixels x spp 32 over the 1.0 M-triangle room, one backward through
all of them.  The oracle does not finish this size in seconds, so the checks are the properties the path offers:
determinism (same draws -> same bits), linearity in the radiance tables (emitter rows + SLF cache: scaling both by 2 is
exact in f32), the gradient lives on emitter rows only and is independent of the radiance values (L is linear), and the
direction-sorted tile kernels of the two tracing stages return the bits of the one-ray-per-thread kernels."""
import argparse
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu

RAYS, SPP, CALLS = 8192, 32, 4


class GpuStub(torch.nn.Module):
    """closed-form stand-in for NGPBRDF (tiny-cuda-nn, third party), evaluated on the GPU: model/brdf.py:243-260's contract"""

    def forward(self, x):
        k = torch.tensor([1.3, 2.1, 0.7], device=x.device); ph = torch.tensor([0.1, 0.5, 0.9], device=x.device)
        return {"albedo": 0.5 + 0.4 * torch.sin(x * k + ph), "roughness": 0.35 + 0.3 * torch.sin(x[:, :1] * 1.7 + x[:, 1:2] * 0.9),
                "metallic": 0.5 + 0.5 * torch.sin(x[:, 2:3] * 2.3)}


@pytest.fixture(scope="module")
def setup():
    import bench
    from iris_amd.model.emitter import SLFEmitterLearn
    from iris_amd.utils.dataset import real_ldr
    from tools import synth
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(scene_seed=1, tris=1_000_000, slf_res=256, layout=0)
    room, slf, emi, scene, emitter0 = bench.build_workload(ns, dev)
    tmp = tempfile.mkdtemp()
    ep, sp = os.path.join(tmp, "emitter.pth"), os.path.join(tmp, "vslf.npz")
    torch.save({"is_emitter": torch.from_numpy(emi["is_emitter"]), "emitter_vertices": torch.from_numpy(emi["emitter_vertices"]),
                "emitter_area": torch.from_numpy(emi["emitter_area"]), "emitter_normal": torch.zeros(len(emi["emitter_area"]), 3),
                "emitter_radiance": torch.from_numpy(emi["emitter_radiance"])}, ep)
    torch.save({"mask": torch.from_numpy(slf["mask"]), "voxel_min": slf["voxel_min"], "voxel_max": slf["voxel_max"], "weight": emitter0.slf.state_dict()}, sp)
    em = SLFEmitterLearn(ep, sp)           # radiance parameter stays on the CPU, as the files are loaded (map_location='cpu'): the path must cope
    H, W = 1080, 1920
    K, c2w = synth.camera(H, W, 0)
    o, d, dx, dy = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, True, device=dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    pick = torch.randint(0, H * W, (RAYS,), generator=g).to(dev)
    return {"dev": dev, "scene": scene, "em": em, "rays": (o[pick], d[pick], dx[pick], dy[pick]), "n_emit": int(emi["is_emitter"].sum())}


def _run(setup, seed_base=0, scale=1.0):
    from iris_amd.utils.path_tracing import path_tracing_single
    em, dev = setup["em"], setup["dev"]
    o, d, dx, dy = setup["rays"]
    mat = GpuStub()
    if scale != 1.0:                      # both radiance tables: L is linear in (emitter radiance, SLF radiance) jointly
        with torch.no_grad():
            em.radiance.mul_(scale); em.slf.radiance.mul_(scale)
    em.radiance.grad = None
    outs, loss = [], 0
    w = torch.linspace(0.5, 1.5, RAYS * 3, device=dev).reshape(RAYS, 3)
    try:
        for c in range(CALLS):
            torch.manual_seed(seed_base + c); torch.cuda.manual_seed(seed_base + c)      # path_tracing_single draws with torch.rand on the device
            L = path_tracing_single(setup["scene"], em, mat, o, d, dx, dy, SPP)
            outs.append(L.detach().clone())
            loss = loss + (L * w).sum()
        loss.backward()
        grad = em.radiance.grad.detach().clone()
    finally:
        if scale != 1.0:
            with torch.no_grad():
                em.radiance.mul_(1.0 / scale); em.slf.radiance.mul_(1.0 / scale)
    return outs, grad


@pytest.mark.timeout(900)
def test_cfg5_properties_at_full_size(setup):
    from iris_amd import _lib as L
    a, ga = _run(setup)
    b, gb = _run(setup)
    for x, y in zip(a, b):
        assert x.shape == (RAYS, 3) and torch.equal(x, y)                       # determinism
    assert torch.isfinite(torch.stack(a)).all() and float(torch.stack(a).abs().sum()) > 0
    # the atomics of the backward scatter make the gradient order-dependent in the last bits only
    assert ga.device == setup["em"].radiance.device and torch.allclose(ga, gb, rtol=1e-5, atol=1e-6)
    # gradient support: radiance has n_face rows, indexed by emitter ordinal (model/emitter.py:158, :201-203): only the first K can receive gradient
    K = setup["n_emit"]
    assert float(ga[K:].abs().sum()) == 0.0 and int((ga[:K].abs().sum(-1) > 0).sum()) > K // 2
    # linearity: radiance x 2 -> L x 2 exactly (power-of-two scaling), gradient unchanged (L is linear in radiance)
    c, gc = _run(setup, scale=2.0)
    for x, y in zip(a, c):
        assert torch.equal(x * 2.0, y)
    assert torch.allclose(ga, gc, rtol=1e-5, atol=1e-6)
    # tiled == untiled tracing stages at this size (262 144 rays per call)
    L.debug_set("pt_tile_min", 1)
    try:
        t, gt = _run(setup)
    finally:
        L.debug_set("pt_tile_min", -1)
    for x, y in zip(a, t):
        assert torch.equal(x, y)
    assert torch.allclose(ga, gt, rtol=1e-5, atol=1e-6)
