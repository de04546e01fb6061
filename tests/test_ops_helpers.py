"""utils/ops.py's small helpers as HIP calls (SURVEY.md section 8 a3 / a4 "helpers"): get_normal_space, angle2xyz, double_sided against the
reference's own outputs (tests/golden/frame.npz) and the oracle; the GGX / Fresnel terms against their formulas evaluated in f32."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def test_frame_angle_double_sided(oracle_mod):
    from iris_amd.utils import ops
    dev = torch.device("cuda:0")
    g = golden("frame.npz")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f = ops.get_normal_space(T(g["normal"]))
    assert f.shape == g["frames"].shape
    np.testing.assert_allclose(f.cpu().numpy(), g["frames"], atol=1e-6, rtol=0)                    # vs the reference
    np.testing.assert_array_equal(f.cpu().numpy(), oracle_mod.get_normal_space(g["normal"]))         # vs the oracle, bit for bit
    assert ops.get_normal_space(T(g["normal"]).reshape(13, -1, 3)).shape == (13, g["normal"].shape[0] // 13, 3, 3)
    xyz = ops.angle2xyz(T(g["theta"]), T(g["phi"]))
    np.testing.assert_allclose(xyz.cpu().numpy(), g["xyz"], atol=1e-6, rtol=0)
    N = T(g["N"]).clone()
    out = ops.double_sided(T(g["V"]), N)
    assert out.data_ptr() == N.data_ptr()                                                          # in place, as the reference
    np.testing.assert_array_equal(N.cpu().numpy(), g["N_flipped"])
    Nt = T(g["N"]).t().contiguous().t()                                                            # a non-contiguous view is updated too
    ops.double_sided(T(g["V"]), Nt)
    np.testing.assert_array_equal(Nt.cpu().numpy(), g["N_flipped"])
    assert ops.get_normal_space(torch.empty(0, 3, device=dev)).shape == (0, 3, 3)


def test_ggx_and_fresnel_terms():
    from iris_amd.utils import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    n = 4096
    c = rng.random(n).astype(np.float32); v = rng.random(n).astype(np.float32); l = rng.random(n).astype(np.float32)
    eta = (0.02 + 0.98 * rng.random(n)).astype(np.float32); f0 = rng.random(n).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(dev)
    f32 = np.float32
    a2 = (eta * eta) * (eta * eta)
    den = c * c * (a2 - f32(1)) + f32(1)
    np.testing.assert_allclose(ops.D_GGX(T(c), T(eta)).cpu().numpy(), a2 / (f32(np.pi) * den * den), rtol=2e-6)
    k = (eta + f32(1)) * (eta + f32(1)) / f32(8)
    g1 = lambda x: f32(1) / (x * (f32(1) - k) + k)
    np.testing.assert_allclose(ops.G1_GGX_Schlick(T(v), T(eta)).cpu().numpy(), g1(v), rtol=2e-6)
    np.testing.assert_allclose(ops.G_Smith(T(v), T(l), T(eta)).cpu().numpy(), g1(l) * g1(v), rtol=4e-6)
    x = (f32(1) - v) ** 5
    np.testing.assert_allclose(ops.fresnelSchlick(T(v), T(f0)).cpu().numpy(), f0 + (f32(1) - f0) * x, rtol=2e-6, atol=1e-7)
    s0, s1 = ops.fresnelSchlick_sep(T(v))
    np.testing.assert_allclose(s0.cpu().numpy(), f32(1) - x, rtol=2e-6, atol=1e-7); np.testing.assert_allclose(s1.cpu().numpy(), x, rtol=4e-6, atol=1e-9)
    # broadcasting as the torch formulas of the reference allow: (B,1) against a scalar roughness, (B,1) VoH against (B,3) F0
    d = ops.D_GGX(T(c).reshape(-1, 1), 0.3)
    assert d.shape == (n, 1)
    np.testing.assert_array_equal(d.reshape(-1).cpu().numpy(), ops.D_GGX(T(c), T(np.full(n, 0.3, np.float32))).cpu().numpy())
    F = ops.fresnelSchlick(T(v).reshape(-1, 1), T(np.stack([f0, f0 * f32(0.5), f0 * f32(0.25)], 1)))
    assert F.shape == (n, 3)
    np.testing.assert_array_equal(F[:, 0].cpu().numpy(), ops.fresnelSchlick(T(v), T(f0)).cpu().numpy())


def test_eval_diffuse_specular_compose_to_eval_brdf():
    """BaseBRDF.eval_diffuse / eval_specular (model/brdf.py:70-110) and the module-level samplers: kd diffuse + ks spec0 + spec1 is eval_brdf,
    whose values the reference pinned (tests/golden/pt_units.npz)."""
    from iris_amd.model.brdf import BaseBRDF, diffuse_sampler, specular_sampler
    dev = torch.device("cuda:0")
    u = golden("pt_units.npz")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    wi, wo, n = T(u["wi"]), T(u["wo"]), T(u["normal"])
    albedo, rough, metal = T(u["albedo"]), T(u["roughness"]), T(u["metallic"])
    b = BaseBRDF()
    bd, pd = b.eval_diffuse(wi, n)
    s0, s1, ps = b.eval_specular(wi, wo, n, rough)
    kd, ks = albedo * (1 - metal), 0.04 * (1 - metal) + albedo * metal
    brdf, pdf = kd * bd + ks * s0 + s1, 0.5 * ps + 0.5 * pd
    rel = lambda a, r: float(np.linalg.norm(a.cpu().numpy().astype(np.float64) - r) / np.linalg.norm(r))
    # (D_GGX's denominator cancels for the peaked lobes among these materials: values up to 9e4 carry the rounding of a different operation order)
    assert rel(brdf, u["brdf"]) <= 5e-5 and rel(pdf, u["brdf_pdf"]) <= 5e-5
    g = golden("sample_diffuse.npz")
    np.testing.assert_array_equal(diffuse_sampler(T(g["u2"]), T(g["normal"])).cpu().numpy(), b.sample_diffuse(T(g["u2"]), T(g["normal"]))[0].cpu().numpy())
    np.testing.assert_allclose(diffuse_sampler(T(g["u2"]), T(g["normal"])).cpu().numpy(), g["wi"], atol=2e-6, rtol=0)
    # one roughness per sample (what sample_brdf hands to specular_sampler): row by row the scalar-roughness result
    B = wo.shape[0]
    u2 = torch.rand(B, 2, device=dev)
    r = torch.where(torch.arange(B, device=dev) % 2 == 0, 0.3, 0.7).reshape(B, 1)
    got = b.sample_specular(u2, wo, n, r)
    for val, rows in ((0.3, slice(0, None, 2)), (0.7, slice(1, None, 2))):
        ref = b.sample_specular(u2, wo, n, val)
        for a, c in zip(got, ref):
            np.testing.assert_array_equal(a[rows].cpu().numpy(), c[rows].cpu().numpy())
    np.testing.assert_array_equal(specular_sampler(u2, r, wo, n).cpu().numpy(), got[0].cpu().numpy())


def test_raw_kernels_refuse_inputs_that_require_grad():
    """the reference's utils/ops.py helpers are differentiable torch ops; these are kernels without a backward pass and must say so instead
    of silently cutting the graph"""
    from iris_amd import _lib as L
    from iris_amd.utils import ops
    dev = torch.device("cuda:0")
    from iris_amd.model.brdf import BaseBRDF
    x = torch.rand(16, device=dev, requires_grad=True)
    n = torch.nn.functional.normalize(torch.randn(16, 3, device=dev), dim=-1)
    for call in (lambda: ops.D_GGX(x, 0.3), lambda: ops.G_Smith(x.detach(), x, 0.3), lambda: ops.fresnelSchlick_sep(x), lambda: ops.angle2xyz(x, x.detach()),
                 lambda: ops.get_normal_space(n.clone().requires_grad_()), lambda: ops.lerp_specular(torch.rand(16, 6, 3, device=dev), x.reshape(-1, 1)),
                 lambda: BaseBRDF().eval_specular(n, n, n, x.reshape(-1, 1))):
        with pytest.raises(L.IrisError, match="backward"):
            call()
    with torch.no_grad():                                         # stating that no gradient is wanted: fine
        assert ops.D_GGX(x, 0.3).shape == (16,)
    assert ops.D_GGX(x.detach(), 0.3).shape == (16,)
