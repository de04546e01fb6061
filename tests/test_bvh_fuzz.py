"""Property test (hypothesis): for random triangle soups -- including degenerate, duplicated, axis-aligned and needle triangles --
and random rays, the GPU BVH traversal (both node layouts) returns exactly the brute-force closest hit of the oracle:
same triangle id (lexicographic (t, id) minimum), same barycentrics, same position bits."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu


@st.composite
def soups(draw):
    seed = draw(st.integers(0, 2 ** 31 - 1))
    n = draw(st.integers(1, 400))
    kind = draw(st.sampled_from(["random", "grid", "needles", "duplicates", "flat"]))
    rng = np.random.default_rng(seed)
    if kind == "grid":                       # axis-aligned wall of shared-edge triangle pairs (ties on edges)
        k = max(1, int(np.sqrt(n / 2)))
        xs, ys = np.meshgrid(np.arange(k + 1), np.arange(k + 1), indexing="ij")
        v = np.stack([xs, ys, np.zeros_like(xs)], -1).reshape(-1, 3).astype(np.float32) / k
        idx = np.arange((k + 1) ** 2).reshape(k + 1, k + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        f = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)]).astype(np.int32)
    else:
        c = rng.uniform(-1, 1, size=(n, 1, 3))
        scale = {"random": 0.3, "needles": 0.3, "duplicates": 0.3, "flat": 0.3}[kind]
        t = c + rng.normal(scale=scale, size=(n, 3, 3))
        if kind == "needles":
            t[:, 2] = t[:, 1] + 1e-4 * rng.normal(size=(n, 3))
        if kind == "flat":
            t[..., 2] = 0.25
        if kind == "duplicates":
            t[n // 2:] = t[: n - n // 2]
        v = t.reshape(-1, 3).astype(np.float32)
        f = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
        if n > 3:
            f[0] = [0, 0, 1]                 # a degenerate triangle
    m = 512
    o = rng.uniform(-1.5, 1.5, size=(m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3)); d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    d[:6] = np.eye(3, dtype=np.float32)[np.arange(6) % 3] * np.where(np.arange(6) % 2, -1, 1)[:, None]
    o[6:40, 2] = 1.0; d[6:40] = [0, 0, -1]   # rays straight down onto z-planes
    return v, f, o, d


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(soups())
def test_gpu_bvh_equals_brute_force(oracle_mod, data):
    from iris_amd import _lib as L
    from iris_amd.utils.path_tracing import Scene, ray_intersect
    v, f, o, d = data
    dev = torch.device("cuda:0")
    osc = oracle_mod.Scene(v, f)
    op, on, ouv, oidx, ovalid = osc.ray_intersect(o, d, brute=True)
    to = torch.from_numpy(o).to(dev); td = torch.from_numpy(d).to(dev)
    for layout in (L.BVH4_Q8, L.BVH4_F32):
        sc = Scene(v, f, device=dev, layout=layout)
        p, n, uv, idx, valid = ray_intersect(sc, to, td)
        np.testing.assert_array_equal(idx.cpu().numpy(), oidx)
        np.testing.assert_array_equal(uv.cpu().numpy(), ouv)
        np.testing.assert_array_equal(p.cpu().numpy(), op)
