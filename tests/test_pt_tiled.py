"""The large-batch variants of the path-tracing stages (pt_tiled_kernel: direction-sorted tiles + persistent-lane traversal) must give
the bits of the one-ray-per-thread kernels; iris_debug_set("pt_tile_min") forces either path.  Also re-runs the refine / path_tracing_single parity
tests with the tile path forced."""
import os

import numpy as np
import pytest
import torch

from test_hip_parity import dev, room_setup, T  # noqa: F401  (fixtures)

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


@pytest.mark.parametrize("N", [1, 100, 129, 4096, 70001])
def test_one_launch_bounce_equals_the_two_stages(tmp_path, N):
    """iris_pt_bounce (trace_indirect's emitter-sampling stage and BRDF stage of a bounce behind ONE launch: 2 N rays sorted and traced together) against the two stage calls,
    every output bit for bit -- through the merged tile kernel (pt_tile_min = 1) and through its small-call fall-back."""
    from test_pt_single import _gpu_setup
    from iris_amd import _lib as L
    dev = torch.device("cuda:0")
    _, _, sc, em = _gpu_setup(tmp_path, dev)
    g = torch.Generator(device="cpu").manual_seed(N)
    R = lambda *s: torch.rand(*s, generator=g).to(dev)
    pos = (R(N, 3) * torch.tensor([3.6, 2.6, 2.2], device=dev) + 0.2).contiguous()
    nrm = torch.nn.functional.normalize(R(N, 3) - 0.5, dim=-1).contiguous()
    wo = torch.nn.functional.normalize(nrm + 0.8 * (R(N, 3) - 0.5), dim=-1).contiguous()
    albedo, rough, metal = R(N, 3).contiguous(), (R(N) * 0.9 + 0.05).contiguous(), R(N).contiguous()
    s1, s2, s1b, s2b = R(N), R(N, 2), R(N), R(N, 2)

    def outs():
        return ([torch.empty(N, 3, device=dev), torch.empty(N, device=dev, dtype=torch.int32)],
                [torch.empty(N, 3, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev),
                 torch.empty(N, device=dev, dtype=torch.int64), torch.empty(N, device=dev, dtype=torch.bool)])
    lib = L.lib()
    for tile_min in (1 << 40, 1):
        L.debug_set("pt_tile_min", tile_min)
        try:
            a1, b1 = outs()
            with torch.cuda.device(dev):
                L.check(lib.iris_pt_nee(sc.handle, em.handle(dev), L.ptr(pos), L.ptr(nrm), L.ptr(wo), L.ptr(albedo), L.ptr(rough), L.ptr(metal), L.ptr(s1), L.ptr(s2), N,
                                        L.ptr(a1[0]), L.ptr(a1[1]), 1e-12, 1e-12, 0.0, L.stream()))
                L.check(lib.iris_pt_brdf_trace(sc.handle, L.ptr(pos), L.ptr(nrm), L.ptr(wo), L.ptr(albedo), L.ptr(rough), L.ptr(metal), L.ptr(s1b), L.ptr(s2b), N,
                                               *[L.ptr(t) for t in b1], 0, 0.0, L.stream()))
                a2, b2 = outs()
                L.check(lib.iris_pt_bounce(sc.handle, em.handle(dev), L.ptr(pos), L.ptr(nrm), L.ptr(wo), L.ptr(albedo), L.ptr(rough), L.ptr(metal), L.ptr(s1), L.ptr(s2), L.ptr(s1b), L.ptr(s2b), N,
                                           L.ptr(a2[0]), L.ptr(a2[1]), 1e-12, 1e-12, 0.0, *[L.ptr(t) for t in b2], L.stream()))
            for k, (x, y) in enumerate(zip(a1 + b1, a2 + b2)):
                assert torch.equal(x, y), (tile_min, k)
        finally:
            L.debug_set("pt_tile_min", -1)
    assert N < 4096 or (int((b1[5] >= 0).sum()) > 0 and int((a1[1] >= 0).sum()) > 0)       # hits and visible emitters are exercised


def test_refine_and_pt_single_parity_through_tiles(tmp_path, oracle_mod, force_tiles):
    import test_refine, test_pt_single
    test_refine.test_hip_refine(tmp_path, oracle_mod)
    fn = getattr(test_pt_single, "test_hip_pt_single", None) or getattr(test_pt_single, "test_hip_path_tracing_single", None)
    if fn is not None:
        import inspect
        kw = {k: v for k, v in (("tmp_path", tmp_path), ("oracle_mod", oracle_mod)) if k in inspect.signature(fn).parameters}
        fn(**kw)


def test_latency_mode_equals_phase_mode(dev, room_setup):
    """Small launches of the one-ray-per-lane kernels run in LATENCY MODE (iris_trace.h trace_q8_joint: the node and the triangle loads of an iteration issued together);
    iris_debug_set('joint_max_rays', 0) forces the phase-scheduled instantiation.  The closest hit does not depend on the schedule: ray_intersect and the two tracing
    stages give the same bits either way, on incoherent rays (random origins on the surface, random directions)."""
    from iris_amd import _lib as L
    from iris_amd.utils.path_tracing import ray_intersect
    s = room_setup
    g = torch.Generator().manual_seed(3)
    n = 40000
    pick = torch.randint(0, len(s["pos"]), (n,), generator=g)
    pos = T(s["pos"], dev)[pick.to(dev)].contiguous(); nrm = T(s["nrm"], dev)[pick.to(dev)].contiguous()
    d = torch.randn(n, 3, generator=g); d = (d / d.norm(dim=1, keepdim=True)).to(dev)
    o = (pos + 1e-3 * nrm).contiguous()
    wo = (-d).contiguous()
    alb = torch.rand(n, 3, generator=g).to(dev); rough = (torch.rand(n, generator=g) * 0.9 + 0.05).to(dev); metal = torch.rand(n, generator=g).to(dev)
    s1, s2 = torch.rand(n, generator=g).to(dev), torch.rand(n, 2, generator=g).to(dev)
    lib = L.lib()

    def stages():
        out = list(ray_intersect(s["sc"], o, d))
        wi = torch.empty(n, 3, device=dev); pdf = torch.empty(n, device=dev); w = torch.empty(n, 3, device=dev); pn = torch.empty(n, 3, device=dev); nn = torch.empty(n, 3, device=dev)
        tri = torch.empty(n, device=dev, dtype=torch.int64); hit = torch.empty(n, device=dev, dtype=torch.bool)
        L.check(lib.iris_pt_brdf_trace(s["sc"].handle, L.ptr(pos), L.ptr(nrm), L.ptr(wo), L.ptr(alb), L.ptr(rough), L.ptr(metal), L.ptr(s1), L.ptr(s2), n,
                                       L.ptr(wi), L.ptr(pdf), L.ptr(w), L.ptr(pn), L.ptr(nn), L.ptr(tri), L.ptr(hit), 0, 0.0, L.stream()))
        coef1 = torch.empty(n, 3, device=dev); e1 = torch.empty(n, device=dev, dtype=torch.int32)
        L.check(lib.iris_pt_nee(s["sc"].handle, s["em"].handle(dev), L.ptr(pos), L.ptr(nrm), L.ptr(wo), L.ptr(alb), L.ptr(rough), L.ptr(metal), L.ptr(s1), L.ptr(s2), n,
                                L.ptr(coef1), L.ptr(e1), 1e-6, 1e-6, 1e-6, L.stream()))
        torch.cuda.synchronize()
        return out + [wi, pdf, w, pn, nn, tri, hit, coef1, e1]
    try:
        L.debug_set("pt_tile_min", 1 << 40)                 # (the one-ray-per-lane kernels, not the tile kernel)
        L.debug_set("joint_max_rays", 1 << 30)
        a = stages()
        L.debug_set("joint_max_rays", 0)
        b = stages()
    finally:
        L.debug_set("joint_max_rays", -1); L.debug_set("pt_tile_min", -1)
    assert int(a[4].sum()) > 0.9 * n                        # a closed room
    for x, y in zip(a, b):
        assert torch.equal(x, y)
