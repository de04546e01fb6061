"""The closest-hit routine is watertight (a2; the reference's intersector is OptiX through Mitsuba's cuda variant, utils/path_tracing.py:30-43,
whose triangle test is watertight): rays aimed EXACTLY at shared vertices, at points on shared edges and a few ulps beside them must hit the
surface.  With the Moeller-Trumbore test of rounds 1-2 a measurable fraction of such rays slipped between two triangles.
Oracle (brute force and BVH) on the CPU; the HIP path against it, bit for bit, on the GPU."""
import numpy as np
import pytest

from conftest import REPO  # noqa: F401


def _patch(n=48, seed=3):
    """an n x n grid over [0,1]^2 at z = 2 with smooth + random displacement (no two triangles coplanar), skirted so that the border is closed"""
    rng = np.random.default_rng(seed)
    s = np.linspace(0.0, 1.0, n + 1)
    X, Y = np.meshgrid(s, s, indexing="ij")
    Z = 2.0 + 0.05 * np.sin(7.1 * X) * np.cos(5.3 * Y) + 0.004 * rng.standard_normal(X.shape)
    v = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    f = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)]).astype(np.int32)
    return v, f, n


def _targets(v, f, n, seed=5):
    """points on the surface where a crack would be: interior vertices, edge midpoints, random points on edges (all in f32), each also nudged by
    a few ulps"""
    rng = np.random.default_rng(seed)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    inner = idx[2:-2, 2:-2].reshape(-1)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    keep = np.isin(e, inner).all(1)
    e = e[keep]
    lam = rng.random(len(e)).astype(np.float32)[:, None]
    pts = [v[inner], (np.float32(0.5) * (v[e[:, 0]] + v[e[:, 1]])).astype(np.float32), (v[e[:, 0]] + lam * (v[e[:, 1]] - v[e[:, 0]])).astype(np.float32)]
    p = np.concatenate(pts)
    nudged = [p]
    for k in (1, 2, 5):
        q = p.copy()
        ax = rng.integers(0, 2, len(p))
        q[np.arange(len(p)), ax] = np.nextafter(q[np.arange(len(p)), ax], np.float32(np.inf) * (rng.integers(0, 2, len(p)) * 2 - 1).astype(np.float32)) if k == 1 else \
            q[np.arange(len(p)), ax] + np.float32(k) * np.spacing(q[np.arange(len(p)), ax]) * (rng.integers(0, 2, len(p)) * 2 - 1).astype(np.float32)
        nudged.append(q.astype(np.float32))
    return np.concatenate(nudged)


def _rays(tg, seed=9):
    rng = np.random.default_rng(seed)
    eyes = np.array([[0.5, 0.5, 0.0], [0.31, 0.77, 0.4], [1.7, -0.9, 0.1], [0.5, 0.5, 4.0]], np.float32)     # below (three) and above the patch
    o = eyes[rng.integers(0, len(eyes), len(tg))]
    d = (tg - o).astype(np.float32)
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def test_oracle_is_watertight_on_edges_and_vertices(oracle_mod):
    v, f, n = _patch()
    o, d = _rays(_targets(v, f, n))
    sc = oracle_mod.Scene(v, f)
    p, nr, uv, idx, valid = sc.ray_intersect(o, d)
    assert valid.all(), f"{(~valid).sum()} of {len(valid)} rays aimed at shared edges / vertices leaked"
    t_bvh = sc.last_t.copy()
    sel = np.arange(0, len(o), 7)
    pb, nb, uvb, idxb, validb = sc.ray_intersect(o[sel], d[sel], brute=True)
    np.testing.assert_array_equal(idxb, idx[sel])            # the BVH is only an accelerator: same lexicographic minimum of (t, index)
    np.testing.assert_array_equal(sc.last_t, t_bvh[sel])
    np.testing.assert_array_equal(pb, p[sel])


def test_oracle_axis_aligned_edges(oracle_mod):
    """exact zeros of the edge functions (the double-precision re-evaluation): an axis-aligned quad grid seen along its grid lines"""
    n = 8
    s = np.arange(n + 1, dtype=np.float32)
    X, Y = np.meshgrid(s, s, indexing="ij")
    v = np.stack([X, Y, np.full_like(X, 3.0)], -1).reshape(-1, 3)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a, b, c, dd = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    f = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, dd], -1).reshape(-1, 3)]).astype(np.int32)
    gx, gy = np.meshgrid(np.arange(1, n, dtype=np.float32), np.arange(1, n, dtype=np.float32), indexing="ij")     # straight at the interior vertices, along -z ... +z
    o = np.stack([gx.reshape(-1), gy.reshape(-1), np.zeros(gx.size, np.float32)], -1)
    d = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (len(o), 1))
    o2 = np.stack([gx.reshape(-1) + 0.5, gy.reshape(-1), np.zeros(gx.size, np.float32)], -1).astype(np.float32)   # edge midpoints
    o3 = np.stack([gx.reshape(-1) - 0.5, gy.reshape(-1) - 0.5, np.zeros(gx.size, np.float32)], -1).astype(np.float32)   # the quads' diagonals
    sc = oracle_mod.Scene(v, f)
    for oo in (o, o2, o3):
        p, nr, uv, ix, valid = sc.ray_intersect(np.ascontiguousarray(oo), d)
        assert valid.all()
        np.testing.assert_array_equal(sc.last_t, np.float32(3.0))
        pb, _, _, ixb, vb = sc.ray_intersect(np.ascontiguousarray(oo), d, brute=True)
        np.testing.assert_array_equal(ixb, ix)                # of the triangles meeting there, the smallest index


@pytest.mark.gpu
def test_hip_is_watertight_and_equals_oracle(oracle_mod):
    import torch
    from iris_amd.utils.path_tracing import Scene, ray_intersect
    dev = torch.device("cuda:0")
    v, f, n = _patch()
    o, d = _rays(_targets(v, f, n))
    sc = Scene(v, f, device=dev)
    pos, nrm, uv, idx, valid = ray_intersect(sc, torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))
    assert bool(valid.all()), f"{int((~valid).sum())} of {len(o)} rays leaked"
    osc = oracle_mod.Scene(v, f)
    p, nr, ouv, oidx, ovalid = osc.ray_intersect(o, d)
    np.testing.assert_array_equal(idx.cpu().numpy(), oidx)
    np.testing.assert_array_equal(pos.cpu().numpy(), p)
    np.testing.assert_array_equal(uv.cpu().numpy(), ouv)


def _icosphere(subdiv=4, seed=11, noise=0.08):
    """closed, consistently indexed triangle mesh: a subdivided icosahedron with radial noise (shared vertices are shared by INDEX and value)"""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        mid, nf = {}, []

        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = v[a] + v[b]; v.append(p / np.linalg.norm(p)); mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    v = np.asarray(v)
    rng = np.random.default_rng(seed)
    r = 1.0 + noise * rng.standard_normal(len(v))
    return (v * r[:, None] * 1.7 + np.array([0.3, -0.2, 0.9])).astype(np.float32), np.asarray(f, np.int32)


def _interior_rays(n, seed=13):
    rng = np.random.default_rng(seed)
    o = (rng.standard_normal((n, 3)) * 0.25 + np.array([0.3, -0.2, 0.9])).astype(np.float32)      # well inside (radius >= 1.7 * (1 - 4 sigma))
    d = rng.standard_normal((n, 3)); d[: n // 8] = np.eye(3)[rng.integers(0, 3, n // 8)] * rng.choice([-1.0, 1.0], (n // 8, 1))   # some axis-parallel
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def test_oracle_closed_mesh_never_leaks(oracle_mod):
    """every ray from inside a closed surface hits it -- the property a watertight test + a conservative BVH guarantee together"""
    v, f = _icosphere()
    o, d = _interior_rays(400_000)
    sc = oracle_mod.Scene(v, f)
    _, _, _, idx, valid = sc.ray_intersect(o, d)
    assert valid.all(), f"{(~valid).sum()} of {len(valid)} rays left a closed mesh"


@pytest.mark.gpu
def test_hip_closed_mesh_never_leaks(oracle_mod):
    import torch
    from iris_amd.utils.path_tracing import Scene, ray_intersect
    dev = torch.device("cuda:0")
    v, f = _icosphere(subdiv=5)                    # 20 480 triangles
    o, d = _interior_rays(4_000_000)
    sc = Scene(v, f, device=dev)
    pos, nrm, uv, idx, valid = ray_intersect(sc, torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))
    assert bool(valid.all()), f"{int((~valid).sum())} of {len(o)} rays left a closed mesh"
    sel = np.arange(0, len(o), 40)
    p, _, ouv, oidx, _ = oracle_mod.Scene(v, f).ray_intersect(o[sel], d[sel])
    np.testing.assert_array_equal(idx.cpu().numpy()[sel], oidx)
    np.testing.assert_array_equal(pos.cpu().numpy()[sel], p)
