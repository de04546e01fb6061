"""The instrumented (`stats`) builds of the bake kernels (include/iris_hip_debug.h) feed bench.py's roofline accounting
(nodes / triangles per ray, lane utilisation, drain share), so their counters are tested like any other output:
invariants that hold by construction, equality of the per-ray quantities between the two kernel variants (how many nodes a ray
visits and how deep its stack gets cannot depend on which rays share a wave), and instrumented outputs == production outputs.
Round 1 shipped a build whose stack-depth counters were garbage ("COUNT build corrupt at HEAD")."""
import numpy as np
import pytest
import torch

from test_hip_parity import dev, room_setup, T  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu

NAMES = ["rays", "node_visits", "tri_tests", "wave_node_iters", "wave_leaf_iters", "sp_gt8", "sp_gt12", "sp_gt16", "tail_sum",
         "drain_node_visits", "drain_wave_node_iters", "top21", "top85", "top341", "top1365", "shared_iters", "shared_lanes", "shared_all_iters", "slow_push_iters", "shared_path_iters"]
PER_RAY = ["rays", "node_visits", "tri_tests", "sp_gt8", "sp_gt12", "sp_gt16", "top21", "top85", "top341", "top1365"]


def _check(st, rays):
    s = dict(zip(NAMES, (int(x) for x in st)))
    assert s["rays"] == rays
    assert 0 <= s["sp_gt16"] <= s["sp_gt12"] <= s["sp_gt8"] <= s["rays"]
    assert s["wave_node_iters"] * 64 >= s["node_visits"] >= s["rays"]          # every ray visits the root
    assert s["wave_leaf_iters"] * 64 >= s["tri_tests"]
    assert s["drain_node_visits"] <= s["node_visits"] and s["drain_wave_node_iters"] <= s["wave_node_iters"]
    assert s["rays"] <= s["top21"] <= s["top85"] <= s["top341"] <= s["top1365"] <= s["node_visits"]
    # node steps in which >= 32 lanes sit at one node of one octant table: at least 32 and at most 64 lanes each, never more steps than there are
    assert s["shared_all_iters"] <= s["shared_iters"] <= s["wave_node_iters"] and 32 * s["shared_iters"] <= s["shared_lanes"] <= 64 * s["shared_iters"]
    assert s["shared_lanes"] <= s["node_visits"]
    assert 0 <= s["slow_push_iters"] <= s["wave_node_iters"]
    # iterations EXECUTED through the scalar path (the instrumented tile kernel follows the timed kernel's wave-level schedule): needs >= 44 lanes at one node, so it is a
    # subset of the ">= 32 lanes at one node" iterations; the pixel-per-wave kernel has no such path
    assert 0 <= s["shared_path_iters"] <= s["shared_iters"]
    return s


@pytest.mark.parametrize("spp", [16, 128])
def test_stats_invariants_and_variant_agreement(dev, room_setup, spp):
    from iris_amd import _lib as L
    from iris_amd import bake_shading as bs
    s = room_setup
    P = min(len(s["pos"]), 3000)
    pos, nrm, wo = T(s["pos"][:P], dev), T(s["nrm"][:P], dev), T(s["wo"][:P], dev)
    for lobe, rough in ((0, None), (1, 0.02), (4, 0.608), (6, 1.0)):
        got = {}
        for variant in (L.BAKE_TILE_SORTED, L.BAKE_PIXEL_PER_WAVE):
            st = torch.zeros(20, device=dev, dtype=torch.int64)
            if rough is None:
                a = (bs.bake_diffuse(s["sc"], s["em"], pos, nrm, spp, seed=1, stream_id=0, stats=st, variant=variant),)
                b = (bs.bake_diffuse(s["sc"], s["em"], pos, nrm, spp, seed=1, stream_id=0, variant=variant),)
            else:
                a = bs.bake_specular(s["sc"], s["em"], pos, nrm, wo, rough, spp, seed=1, stream_id=lobe, stats=st, variant=variant)
                b = bs.bake_specular(s["sc"], s["em"], pos, nrm, wo, rough, spp, seed=1, stream_id=lobe, variant=variant)
            for x, y in zip(a, b):
                assert torch.equal(x, y), "instrumented build must return the production build's bits"
            got[variant] = _check(st.cpu().numpy(), P * spp)
        for k in PER_RAY:
            assert got[L.BAKE_TILE_SORTED][k] == got[L.BAKE_PIXEL_PER_WAVE][k], (lobe, k, got)
        assert got[L.BAKE_PIXEL_PER_WAVE]["drain_node_visits"] == 0 and got[L.BAKE_PIXEL_PER_WAVE]["shared_path_iters"] == 0
        if spp == 128 and lobe in (1, 4):
            assert got[L.BAKE_TILE_SORTED]["shared_path_iters"] > 0           # coherent lobes: the scalar path IS taken in the instrumented build
