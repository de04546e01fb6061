"""Property test (hypothesis): for random triangle soups -- including degenerate, duplicated, axis-aligned, needle and split long triangles --
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
    kind = draw(st.sampled_from(["random", "grid", "needles", "duplicates", "flat", "long"]))
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
        scale = {"random": 0.3, "needles": 0.3, "duplicates": 0.3, "flat": 0.3, "long": 0.02}[kind]
        t = c + rng.normal(scale=scale, size=(n, 3, 3))
        if kind == "long":                   # a few triangles far longer than the rest: referenced through several clipped boxes (presplit)
            k = max(1, n // 50)
            t[:k] = rng.uniform(-1.2, 1.2, size=(k, 3, 3))
            if k > 1:
                t[1, :, 2] = -0.5            # one of them axis-aligned (a floor)
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


def test_long_triangles_are_split_and_hits_do_not_change(oracle_mod):
    """Early split clipping (bvh_build.cpp): long triangles among fine ones get several leaf records; hits are those of the unsplit tree
    and of brute force, bit for bit, and the traversal visits fewer nodes."""
    from iris_amd import _lib as L
    from iris_amd.utils.path_tracing import Scene, ray_intersect
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    n = 20000
    c = rng.uniform(0, 4, size=(n, 1, 3))
    t = c + rng.normal(scale=0.02, size=(n, 3, 3))
    walls = np.array([[[0, 0, 0], [4, 0, 0], [4, 4, 0]], [[0, 0, 0], [4, 4, 0], [0, 4, 0]], [[0, 0, 0], [0, 4, 4], [0, 0, 4]],
                      [[0.5, 0.2, 3.9], [3.7, 3.1, 0.1], [3.9, 3.3, 0.2]]], np.float64)
    t = np.concatenate([t, walls])
    v = t.reshape(-1, 3).astype(np.float32); f = np.arange(3 * len(t), dtype=np.int32).reshape(-1, 3)
    m = 65536
    o = rng.uniform(0.2, 3.8, size=(m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3)); d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    out = {}
    for presplit in (0, 80):
        L.debug_set("bvh_presplit_x10", presplit)
        try:
            sc = Scene(v, f, device=dev)
        finally:
            L.debug_set("bvh_presplit_x10", -1)
        info = sc.info()
        assert info["n_triangles"] == len(f)
        out[presplit] = (info["n_leaf_records"], info["sah_cost"], [x.cpu().numpy() for x in ray_intersect(sc, to, td)])
    assert out[0][0] == len(f) and len(f) + 100 < out[80][0] <= len(f) + max(len(f) // 4, 1024)
    assert out[80][1] < out[0][1]                                            # SAH cost of the tree
    for a, b in zip(out[0][2], out[80][2]):
        np.testing.assert_array_equal(a, b)
    osc = oracle_mod.Scene(v, f)
    op, on, ouv, oidx, ovalid = osc.ray_intersect(o[:4096], d[:4096], brute=True)
    np.testing.assert_array_equal(out[80][2][3][:4096], oidx)
    np.testing.assert_array_equal(out[80][2][0][:4096], op)
    assert (out[80][2][3] >= n).mean() > 0.2                                 # the long triangles are what most rays end on
