"""BASELINE cfg 5 / SURVEY.md section 8 a9: path_tracing_single (utils/path_tracing.py:320-407) and its building blocks
sample_emitter / eval_brdf / sample_brdf, forward and d/d(emitter.radiance).

Goldens (tests/golden/pt_units.npz, pt_single.npz) come from the reference's own Python (tools/make_goldens.py) with a
closed-form stub material (the NGP hash grid is third party) and the oracle's closest hit patched in for Mitsuba."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_l2
from stub_material import StubMaterial, stub_material_np


@pytest.fixture(params=[0, 1], ids=["libm", "device-arithmetic"])
def omode(request, oracle_mod):
    oracle_mod.set_mode(request.param)
    yield request.param
    oracle_mod.set_mode(0)


def _box(oracle_mod, with_sampling=True):
    g, p = golden("bake_box.npz"), golden("pt_single.npz")
    sc = oracle_mod.Scene(g["verts"], g["faces"])
    slf = oracle_mod.VoxelSLF(g["slf_inds"], g["slf_radiance"], float(g["voxel_min"]), float(g["voxel_max"]))
    em = oracle_mod.SLFEmitter(g["is_emitter"], g["emitter_radiance"], g["emitter_area"], slf, p["emitter_vertices"], p["emitter_cdf"])
    return g, p, sc, em


def check_units(got_se, got_eb, got_sb, u):
    wi, pdf, tri = got_se
    np.testing.assert_array_equal(tri, u["se_tri"])
    np.testing.assert_allclose(wi, u["se_wi"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(pdf, u["se_pdf"], rtol=1e-6)
    brdf, bpdf = got_eb
    assert rel_l2(brdf, u["brdf"]) <= 1e-5 and rel_l2(bpdf, u["brdf_pdf"]) <= 1e-5
    np.testing.assert_allclose(brdf, u["brdf"], rtol=2e-4, atol=1e-5)
    swi, spdf, sw = got_sb
    np.testing.assert_allclose(swi, u["sb_wi"], atol=8e-6, rtol=0)
    up = (u["wo"] * u["normal"]).sum(-1) > 0.02                      # the path only produces wo above the surface
    assert rel_l2(sw[up], u["sb_weight"][up]) <= 1e-5                 # brdf/pdf: the GGX D cancels -> well conditioned
    # the pdf itself carries D_GGX, whose denominator NoH^2(a^2-1)+1 cancels for peaked lobes (values up to 9e4 here)
    ok = np.abs(spdf - u["sb_pdf"])[up] <= 1e-3 * np.abs(u["sb_pdf"][up]) + 1e-5
    assert ok.mean() >= 0.98 and rel_l2(spdf[up], u["sb_pdf"][up]) <= 1e-2


def test_oracle_units(oracle_mod, omode):
    _, _, _, em = _box(oracle_mod)
    u = golden("pt_units.npz")
    mat = {"albedo": u["albedo"], "roughness": u["roughness"], "metallic": u["metallic"]}
    check_units(em.sample_emitter(u["s1"], u["s2"], u["position"]), oracle_mod.eval_brdf(u["wi"], u["wo"], u["normal"], mat),
                oracle_mod.sample_brdf(u["sb_s1"], u["sb_s2"], u["wo"], u["normal"], mat), u)


def test_oracle_path_tracing_single(oracle_mod, omode):
    g, p, sc, em = _box(oracle_mod)
    L, terms = oracle_mod.path_tracing_single(sc, em, stub_material_np, p["rays_o"], p["rays_d"], p["dx_du"], p["dy_dv"], int(p["spp"]),
                                              [p[f"u{k}"] for k in range(5)], radiance=p["radiance"])
    assert len(terms["e1"]) == p["u1"].shape[0]                       # same number of surviving paths as the reference drew for
    assert rel_l2(L, p["L"]) <= 1e-5
    gr = oracle_mod.grad_radiance(terms, p["grad_weight"], p["radiance"].shape[0])
    assert rel_l2(gr, p["grad_radiance"]) <= 1e-5
    assert (np.abs(p["grad_radiance"]).sum(-1) > 0).sum() >= 1


# --------------------------------------------------------------------------------------------------------- GPU
def _gpu_setup(tmp_path, dev):
    from iris_amd.model.emitter import SLFEmitterLearn
    from iris_amd.model.slf import VoxelSLF
    from iris_amd.utils.path_tracing import Scene
    g, p = golden("bake_box.npz"), golden("pt_single.npz")
    slf = VoxelSLF(torch.from_numpy(g["slf_mask"]), float(g["voxel_min"]), float(g["voxel_max"]))
    slf.radiance[:] = torch.from_numpy(g["slf_radiance"])
    ep, sp = str(tmp_path / "emitter.pth"), str(tmp_path / "vslf.npz")
    torch.save({"is_emitter": torch.from_numpy(g["is_emitter"]), "emitter_vertices": torch.from_numpy(p["emitter_vertices"]),
                "emitter_area": torch.from_numpy(g["emitter_area"]), "emitter_normal": torch.zeros(2, 3),
                "emitter_radiance": torch.from_numpy(p["radiance"])}, ep)
    torch.save({"mask": torch.from_numpy(g["slf_mask"]), "voxel_min": float(g["voxel_min"]), "voxel_max": float(g["voxel_max"]), "weight": slf.state_dict()}, sp)
    em = SLFEmitterLearn(ep, sp).to(dev)
    return g, p, Scene(g["verts"], g["faces"], device=dev), em


@pytest.mark.gpu
def test_hip_units(tmp_path, oracle_mod):
    from iris_amd.model.brdf import BaseBRDF
    dev = torch.device("cuda:0")
    _, _, _, em = _gpu_setup(tmp_path, dev)
    u = golden("pt_units.npz")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    N = lambda t: t.detach().cpu().numpy()
    mat = {"albedo": T(u["albedo"]), "roughness": T(u["roughness"]), "metallic": T(u["metallic"])}
    np.testing.assert_array_equal(N(em.emitter_cdf), u["emitter_cdf"])      # torch reproduces the reference's cdf buffer
    se = tuple(N(t) for t in em.sample_emitter(T(u["s1"]), T(u["s2"]), T(u["position"])))
    eb = tuple(N(t) for t in BaseBRDF().eval_brdf(T(u["wi"]), T(u["wo"]), T(u["normal"]), mat))
    sb = tuple(N(t) for t in BaseBRDF().sample_brdf(T(u["sb_s1"]), T(u["sb_s2"]), T(u["wo"]), T(u["normal"]), mat))
    check_units(se, eb, sb, u)
    # bit for bit against the device-arithmetic oracle
    _, _, _, oem = _box(oracle_mod)
    omat = {"albedo": u["albedo"], "roughness": u["roughness"], "metallic": u["metallic"]}
    with oracle_mod.device_arithmetic():
        ose = oem.sample_emitter(u["s1"], u["s2"], u["position"])
        oeb = oracle_mod.eval_brdf(u["wi"], u["wo"], u["normal"], omat)
        osb = oracle_mod.sample_brdf(u["sb_s1"], u["sb_s2"], u["wo"], u["normal"], omat)
    for a, b in zip(se + eb + sb, ose + oeb + osb):
        np.testing.assert_array_equal(a.reshape(b.shape), b)


@pytest.mark.gpu
def test_hip_path_tracing_single_forward_backward(tmp_path, oracle_mod):
    from iris_amd.utils.path_tracing import path_tracing_single
    dev = torch.device("cuda:0")
    g, p, sc, em = _gpu_setup(tmp_path, dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    unif = [T(p[f"u{k}"]) for k in range(5)]
    L = path_tracing_single(sc, em, StubMaterial(), T(p["rays_o"]), T(p["rays_d"]), T(p["dx_du"]), T(p["dy_dv"]), int(p["spp"]), uniforms=unif)
    assert L.shape == (p["rays_o"].shape[0], 3) and L.requires_grad
    assert rel_l2(L.detach().cpu().numpy(), p["L"]) <= 1e-5                  # vs the reference
    (gr,) = torch.autograd.grad((L * T(p["grad_weight"])).sum(), em.radiance)
    assert rel_l2(gr.cpu().numpy(), p["grad_radiance"]) <= 1e-5             # vs the reference's autograd
    # forward bit for bit against the device-arithmetic oracle (same staging, same material inputs)
    _, _, osc, oem = _box(oracle_mod)
    with oracle_mod.device_arithmetic():
        oL, terms = oracle_mod.path_tracing_single(osc, oem, stub_material_np, p["rays_o"], p["rays_d"], p["dx_du"], p["dy_dv"], int(p["spp"]),
                                                   [p[f"u{k}"] for k in range(5)], radiance=p["radiance"])
    np.testing.assert_array_equal(L.detach().cpu().numpy(), oL)
    # the un-compacted mode (no host synchronisation: what a training loop runs): with the reference's draws moved to their rays' indices it is
    # the same arithmetic in the same order -- the forward pass bit for bit
    from iris_amd import _lib as Lb
    from iris_amd.utils.path_tracing import ray_intersect
    B, spp = p["rays_o"].shape[0], int(p["spp"])
    wi0 = torch.empty(B * spp, 3, device=dev)
    rd, dxu, dyv, ro, dudv = T(p["rays_d"]), T(p["dx_du"]), T(p["dy_dv"]), T(p["rays_o"]), unif[0].reshape(2, B, spp).contiguous()      # (kept alive across the launch)
    Lb.check(Lb.lib().iris_pt_jitter(Lb.ptr(rd), Lb.ptr(dxu), Lb.ptr(dyv), Lb.ptr(dudv), B, spp, Lb.ptr(wi0), Lb.stream()))
    _, _, _, tri0, _ = ray_intersect(sc, ro.repeat_interleave(spp, 0), wi0)
    e0 = torch.empty(B * spp, device=dev, dtype=torch.int32); vn = torch.empty(B * spp, device=dev, dtype=torch.bool)
    Lb.check(Lb.lib().iris_pt_primary_emit(em.handle(dev), Lb.ptr(tri0), B * spp, Lb.ptr(e0), Lb.ptr(vn), Lb.stream()))
    assert int(vn.sum()) == unif[1].numel() and 0 < int(vn.sum()) < B * spp
    wide = [unif[0]]
    for k in (1, 2, 3, 4):
        w = torch.full((B * spp,) + tuple(unif[k].shape[1:]), 0.25, device=dev); w[vn] = unif[k]; wide.append(w)
    Lm = path_tracing_single(sc, em, StubMaterial(), T(p["rays_o"]), T(p["rays_d"]), T(p["dx_du"]), T(p["dy_dv"]), spp, uniforms=wide, compact=False)
    assert torch.equal(Lm.detach(), L.detach())
    (gm,) = torch.autograd.grad((Lm * T(p["grad_weight"])).sum(), em.radiance)
    assert rel_l2(gm.cpu().numpy(), gr.cpu().numpy()) <= 1e-5                 # (a scatter of float atomics: equal up to summation order)
    # random draws path (no uniforms given): finite, deterministic shape, gradient reaches only emitter rows
    torch.manual_seed(0)
    L2 = path_tracing_single(sc, em, StubMaterial(), T(p["rays_o"]), T(p["rays_d"]), T(p["dx_du"]), T(p["dy_dv"]), 8)
    (g2,) = torch.autograd.grad(L2.sum(), em.radiance)
    assert torch.isfinite(L2).all() and torch.isfinite(g2).all()
    assert int((g2.abs().sum(-1) > 0).sum()) <= int(g["is_emitter"].sum())


@pytest.mark.gpu
def test_skipping_the_unused_material_evaluation_changes_nothing(tmp_path):
    """The reference evaluates the material network a second time at the sampled hits (utils/path_tracing.py:392) and uses only `roughness > trace_roughness = 0.0`
    of it (model/emitter.py:209).  NGPBRDF's roughness is sigmoid * 0.98 + 0.02 >= 0.02, so path_tracing_single does not launch that evaluation
    (skip_unused_material, the default): L and the gradient must be the bits of the literal evaluation order -- for the function's own draws (one fused
    generator launch) and for recorded draws in the compacted mode; a callable without a declared bound is always evaluated."""
    from iris_amd.model.brdf import NGPBRDF
    from iris_amd.utils.path_tracing import path_tracing_single
    dev = torch.device("cuda:0")
    g, p, scene, em = _gpu_setup(tmp_path, dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)          # noqa: E731
    mat = StubMaterial()
    net = NGPBRDF(float(g["voxel_min"]), float(g["voxel_max"]))
    net.load_state_dict({"mlp.params": (torch.rand(net.mlp.params.numel(), generator=torch.Generator().manual_seed(2)) * 2 - 1) * 0.3})
    assert net.roughness_min == 0.02
    rays = [T(p[k]) for k in ("rays_o", "rays_d", "dx_du", "dy_dv")]
    spp = 16
    calls = []
    orig = NGPBRDF.forward

    def counting(self, position):
        calls.append(position.shape[0])
        out = orig(self, position)
        assert float(out["roughness"].min()) >= 0.02               # the bound the skip relies on
        return out
    NGPBRDF.forward = counting
    try:
        res = {}
        for skip in (True, False):
            calls.clear()
            em.radiance.grad = None
            torch.manual_seed(5); torch.cuda.manual_seed(5)
            L = path_tracing_single(scene, em, net, *rays, spp, skip_unused_material=skip)
            L.square().sum().backward()
            res[skip] = (L.detach().clone(), em.radiance.grad.clone(), len(calls))
        assert res[True][2] == 1 and res[False][2] == 2
        assert torch.equal(res[True][0], res[False][0])
        assert rel_l2(res[True][1].cpu().numpy(), res[False][1].cpu().numpy()) <= 2e-5           # (the backward pass scatters ~10^5 float atomics into the box room's TWO emitter rows: equal up to summation order, skip or no skip; two runs of ONE mode differ as much)
        assert float(res[True][0].abs().sum()) > 0 and float(res[True][1].abs().sum()) > 0
        # recorded draws (the reference's compacted mode): the same
        u = [T(p[f"u{k}"]) for k in range(5)]
        La = path_tracing_single(scene, em, net, *rays, int(p["spp"]), uniforms=u, skip_unused_material=True)
        Lb = path_tracing_single(scene, em, net, *rays, int(p["spp"]), uniforms=u, skip_unused_material=False)
        assert torch.equal(La, Lb)
        # a callable that declares nothing is evaluated twice whatever the flag
        seen = []

        def plain(position):
            seen.append(1)
            return mat(position)
        path_tracing_single(scene, em, plain, *rays, spp, skip_unused_material=True)
        assert len(seen) == 2
    finally:
        NGPBRDF.forward = orig


@pytest.mark.gpu
def test_training_step_replayed_as_hip_graph(tmp_path):
    """forward + backward of the un-compacted mode captured once as a HIP graph (torch.cuda.CUDAGraph) and replayed: every launch of the path
    is stream-ordered and allocates nothing outside torch's pool, so the replay gives what the eager step gives (the scatter-add of the
    backward pass: up to summation order)."""
    from iris_amd.utils.path_tracing import path_tracing_single
    from tools.bench_pt_single import GpuStub
    dev = torch.device("cuda:0")
    g, p, sc, em = _gpu_setup(tmp_path, dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    B, spp = p["rays_o"].shape[0], int(p["spp"])
    gen = torch.Generator(device="cpu").manual_seed(3)
    unif = [torch.rand(2, B, spp, 1, generator=gen).to(dev)] + [torch.rand(*((B * spp,) + t), generator=gen).to(dev) for t in ((), (2,), (), (2,))]
    ro, rd, dxu, dyv, w = T(p["rays_o"]), T(p["rays_d"]), T(p["dx_du"]), T(p["dy_dv"]), T(p["grad_weight"])
    mat = GpuStub()

    def step():
        L = path_tracing_single(sc, em, mat, ro, rd, dxu, dyv, spp, uniforms=unif, compact=False)
        (L * w).sum().backward()
        return L
    em.radiance.grad = torch.zeros_like(em.radiance)
    L_eager = step().detach().clone(); g_eager = em.radiance.grad.clone()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # (torch's recipe: warm up on a side stream before capturing)
        em.radiance.grad.zero_(); step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        em.radiance.grad.zero_()
        L_static = step()
    for _ in range(2):
        L_static.detach().fill_(-1.0); em.radiance.grad.fill_(-1.0)
        graph.replay(); torch.cuda.synchronize()
        assert torch.equal(L_static.detach(), L_eager)
        assert rel_l2(em.radiance.grad.cpu().numpy(), g_eager.cpu().numpy()) <= 1e-5      # (float atomics: the order of a row's ~10^4 terms differs from run to run; 60 replays: up to 9.8e-7)
    em.radiance.grad = None


# --------------------------------------------------------------------------------------------------------- path_tracing (render.py's integrator)
def _full(oracle_mod):
    g, p = golden("bake_box.npz"), golden("pt_full.npz")
    sc = oracle_mod.Scene(g["verts"], g["faces"])
    slf = oracle_mod.VoxelSLF(g["slf_inds"], g["slf_radiance"], float(g["voxel_min"]), float(g["voxel_max"]))
    em = oracle_mod.SLFEmitter(g["is_emitter"], p["radiance"], g["emitter_area"], slf, p["emitter_vertices"], p["emitter_cdf"])
    return g, p, sc, em, [p[f"u_{k}"] for k in range(int(p["n_u"]))]


def test_oracle_path_tracing_full(oracle_mod, omode):
    """utils/path_tracing.py:214-318 against the reference's own output (tests/golden/pt_full.npz: 17 recorded draws, indir_depth 3):
    the first bounce alone (the reference run with indir_depth 0 on the same draws) and with the continuation."""
    g, p, sc, em, us = _full(oracle_mod)
    args = (sc, em, stub_material_np, p["rays_o"], p["rays_d"], p["dx_du"], p["dy_dv"], int(p["spp"]))
    L0, t0 = oracle_mod.path_tracing_single(*args, us[:5], trace_roughness=0.6)
    assert rel_l2(L0, p["L_first_bounce"]) <= 1e-5
    assert int(t0["valid_next"].sum()) == us[5].shape[0]                 # the paths the reference continued
    L, _ = oracle_mod.path_tracing(*args, int(p["indir_depth"]), us)
    assert rel_l2(L, p["L"]) <= 1e-5
    assert rel_l2(p["L"], p["L_first_bounce"]) > 1e-2                    # (the continuation is not a rounding error of the total)


@pytest.mark.gpu
def test_hip_path_tracing_full(tmp_path, oracle_mod):
    from iris_amd.utils.path_tracing import path_tracing
    dev = torch.device("cuda:0")
    g, _, sc, em = _gpu_setup(tmp_path, dev)
    _, p, osc, oem, us = _full(oracle_mod)
    assert np.array_equal(em.radiance.detach().cpu().numpy(), p["radiance"])
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rays = (T(p["rays_o"]), T(p["rays_d"]), T(p["dx_du"]), T(p["dy_dv"]))
    L = path_tracing(sc, em, StubMaterial(), *rays, int(p["spp"]), int(p["indir_depth"]), uniforms=[T(u) for u in us])
    assert L.shape == (p["rays_o"].shape[0], 3) and L.requires_grad
    assert rel_l2(L.detach().cpu().numpy(), p["L"]) <= 1e-5                  # vs the reference
    L0 = path_tracing(sc, em, StubMaterial(), *rays, int(p["spp"]), 0, uniforms=[T(u) for u in us[:5]])
    assert rel_l2(L0.detach().cpu().numpy(), p["L_first_bounce"]) <= 1e-5
    with oracle_mod.device_arithmetic():                                     # bit for bit against the device-arithmetic oracle
        oL, terms = oracle_mod.path_tracing(osc, oem, stub_material_np, p["rays_o"], p["rays_d"], p["dx_du"], p["dy_dv"], int(p["spp"]), int(p["indir_depth"]), us)
    np.testing.assert_array_equal(L.detach().cpu().numpy(), oL)
    # gradient: through the first bounce only (the continuation rides on the constant term), i.e. the analytic gradient of the oracle's terms
    w = torch.rand(L.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    (gr,) = torch.autograd.grad((L * w).sum(), em.radiance)
    ogr = oracle_mod.grad_radiance(terms, w.cpu().numpy(), p["radiance"].shape[0])
    assert rel_l2(gr.cpu().numpy(), ogr) <= 1e-5 and int((gr.abs().sum(-1) > 0).sum()) >= 1
    # random draws: finite, and deeper continuation only adds light
    torch.manual_seed(0)
    Lr = path_tracing(sc, em, StubMaterial(), *rays, 16, 3)
    torch.manual_seed(0)
    Lr0 = path_tracing(sc, em, StubMaterial(), *rays, 16, 0)
    assert torch.isfinite(Lr).all() and Lr.shape == Lr0.shape and float(Lr.detach().mean()) > float(Lr0.detach().mean())
