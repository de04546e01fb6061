"""The large-batch variants of the path-tracing stages (pt_tiled_kernel: direction-sorted tiles + persistent-lane traversal) must give
the bits of the one-ray-per-thread kernels; iris_debug_set("pt_tile_min") forces either path.  Also re-runs the refine / path_tracing_single parity
tests with the tile path forced."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_tiles():
    from iris_amd import _lib as L
    L.debug_set("pt_tile_min", 1)
    yield
    L.debug_set("pt_tile_min", -1)


def _stage_outputs(scene, em, dev, N, seed):
    from iris_amd import _lib as L
    from iris_amd.utils.path_tracing import _lobe_trace
    g = torch.Generator(device="cpu").manual_seed(seed)
    R = lambda *s: torch.rand(*s, generator=g).to(dev)
    pos = (R(N, 3) * torch.tensor([3.6, 2.6, 2.2], device=dev) + 0.2).contiguous()
    nrm = torch.nn.functional.normalize(R(N, 3) - 0.5, dim=-1).contiguous()
    wo = torch.nn.functional.normalize(nrm + 0.8 * (R(N, 3) - 0.5), dim=-1).contiguous()
    albedo, rough, metal = R(N, 3).contiguous(), (R(N) * 0.9 + 0.05).contiguous(), R(N).contiguous()
    s1, s2 = R(N), R(N, 2)
    out = []
    for lobe, r in ((0, 0.0), (1, 0.0), (2, 0.412)):
        out += list(_lobe_trace(scene, pos, nrm, wo, (albedo, rough, metal), s1, s2, lobe, r))
    coef1 = torch.empty(N, 3, device=dev); e1 = torch.empty(N, device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        L.check(L.lib().iris_pt_nee(scene.handle, em.handle(dev), L.ptr(pos), L.ptr(nrm), L.ptr(wo), L.ptr(albedo), L.ptr(rough), L.ptr(metal), L.ptr(s1), L.ptr(s2), N,
                                    L.ptr(coef1), L.ptr(e1), 1e-12, 1e-12, 0.0, L.stream()))
    return out + [coef1, e1]


@pytest.mark.parametrize("N", [1, 255, 4096, 70001])
def test_tiled_stages_equal_plain_kernels(tmp_path, N):
    from test_pt_single import _gpu_setup
    dev = torch.device("cuda:0")
    _, _, sc, em = _gpu_setup(tmp_path, dev)
    from iris_amd import _lib as L
    L.debug_set("pt_tile_min", 1 << 40)
    try:
        plain = _stage_outputs(sc, em, dev, N, seed=N)
        L.debug_set("pt_tile_min", 1)
        tiled = _stage_outputs(sc, em, dev, N, seed=N)
    finally:
        L.debug_set("pt_tile_min", -1)
    assert len(plain) == len(tiled) == 20
    for k, (a, b) in enumerate(zip(plain, tiled)):
        assert torch.equal(a, b), k
    assert int((plain[5] >= 0).sum()) > 0 or N == 1            # next-hit triangle ids are exercised


def test_refine_and_pt_single_parity_through_tiles(tmp_path, oracle_mod, force_tiles):
    import test_refine, test_pt_single
    test_refine.test_hip_refine(tmp_path, oracle_mod)
    fn = getattr(test_pt_single, "test_hip_pt_single", None) or getattr(test_pt_single, "test_hip_path_tracing_single", None)
    if fn is not None:
        import inspect
        kw = {k: v for k, v in (("tmp_path", tmp_path), ("oracle_mod", oracle_mod)) if k in inspect.signature(fn).parameters}
        fn(**kw)
