"""CPU-side checks of what bench.py's roofline is computed from: the committed counter profiles and how one is chosen (no GPU call)."""
import os
import sys

from conftest import REPO


def test_pmc_profile_is_chosen_by_launch_size():
    """bench.py's roofline reads the committed counters of the launch it timed: the whole view at N = 1, rank 0's stripes of an N-rank run otherwise (profiles taken with
    --emulate-world N on one GPU and labelled so); counters of other kernel sources or of another launch size are refused with the reason.  (CPU: no GPU call.)"""
    sys.path.insert(0, REPO)
    import bench
    full = 1920 * 1080 * 128 * 7
    sizes = {1: full, 2: 68 * 8 * 1920 * 128 * 7, 4: 34 * 8 * 1920 * 128 * 7, 8: 17 * 8 * 1920 * 128 * 7}       # rank 0 owns 135 / 68 / 34 / 17 of the 135 stripes of 8 rows
    for world, rays in sizes.items():
        pj, src = bench.load_pmc(float(rays), 64)
        if pj is None:
            assert "stale" in src, src                      # the committed profiles are of other kernel sources: every one of them refused, with the reason
            continue
        assert pj["rays_per_launch"] == rays and pj["_file"] == ("pmc_r6.json" if world == 1 else f"pmc_r6_world{world}.json")
        assert (pj.get("emulated_world") or 1) == world and (("emulate-world %d" % world) in src) == (world > 1)
        assert pj["counters"]["SQ_INSTS_VALU"] / rays > 50 and (pj["counters"]["FETCH_SIZE"] + pj["counters"]["WRITE_SIZE"]) * 1024 / rays > 100
    pj, src = bench.load_pmc(float(full) * 0.77, 64)        # a launch size nobody profiled
    assert pj is None and ("rays per launch" in src or "stale" in src)
