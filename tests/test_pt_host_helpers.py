"""Host-side helpers of the path-tracing integrators (iris_amd/utils/path_tracing.py) that need no GPU: the one-allocation pool a bounce carves its arrays out of and the
one-launch draws of a bounce."""
import pytest
import torch


def test_pool_pieces_are_aligned_disjoint_views():
    from iris_amd import _lib as L
    from iris_amd.utils.path_tracing import _Pool
    N = 1001
    pool = _Pool(28 * N + 256, "cpu")
    pieces = [pool.f(N, 3), pool.i32(N), pool.f(N, 3), pool.f(N), pool.f(N, 3), pool.f(N, 3), pool.f(N, 3), pool.i64(N), pool.u8(N), pool.f(N, 3), pool.f(N, 3), pool.i32(N), pool.u8(N)]
    spans = []
    for t in pieces:
        assert t.is_contiguous() and t.data_ptr() % 16 == 0 and t.shape[0] == N
        spans.append((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()))
    spans.sort()
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0                                            # no two pieces overlap
    assert pieces[7].dtype == torch.int64 and pieces[8].dtype == torch.bool and pieces[1].dtype == torch.int32
    for k, t in enumerate(pieces):                                 # writes to one piece do not show in another
        t.fill_(1 if t.dtype == torch.bool else k + 1)
    for k, t in enumerate(pieces):
        assert bool((t == (1 if t.dtype == torch.bool else k + 1)).all())
    with pytest.raises(L.IrisError):
        _Pool(10, "cpu").f(1000, 3)                                # an undersized block is an error, not a silently short view


def test_bounce_draws_are_four_aligned_pieces_of_one_launch():
    from iris_amd.utils.path_tracing import _bounce_draws
    for N in (1, 5, 4096, 70001):
        torch.manual_seed(7)
        s1, s2, s1b, s2b = _bounce_draws(None, True, N, "cpu")
        assert s1.shape == (N,) and s2.shape == (N, 2) and s1b.shape == (N,) and s2b.shape == (N, 2)
        for t in (s1, s2, s1b, s2b):
            assert t.is_contiguous() and t.data_ptr() % 16 == 0 and float(t.min()) >= 0.0 and float(t.max()) < 1.0
        base = s1.data_ptr()
        ends = [(t.data_ptr() - base) // 4 + t.numel() for t in (s1, s2, s1b, s2b)]
        starts = [(t.data_ptr() - base) // 4 for t in (s1, s2, s1b, s2b)]
        assert all(e <= s for e, s in zip(ends, starts[1:]))       # disjoint, in order
    # recorded draws: handed out in the reference's order, untouched
    rec = [torch.full((3,), 0.1), torch.full((3, 2), 0.2), torch.full((3,), 0.3), torch.full((3, 2), 0.4)]
    it = iter(rec)
    got = _bounce_draws(lambda *shape: next(it).reshape(*shape), False, 3, "cpu")
    assert all(torch.equal(a, b) for a, b in zip(got, rec))
