"""Parity against the REFERENCE'S OWN PYTHON at room scale (tests/golden/bake_room.npz, tools/make_room_golden.py): the bake loop
bodies bake_shading.py:108-123 / :168-188 replayed through the imported reference functions on torch-CPU on the 200 k-triangle room,
160 x 120 pixels, spp 64, all 13 maps, uniforms = the Philox stream the HIP kernels generate themselves.

north_star asks for <= 1e-4 relative L2 against the reference path.  The integrand is discontinuous (emitter edges, voxel edges, triangle
edges), so two correct evaluations of the reference's formulas that differ in the last bit of a sampled direction disagree on a few
samples per million -- and a single sample that crosses an emitter edge is already > 1e-4 of a map at this size.  The fixture therefore
also carries a per-pixel hash of the reference's per-sample (hit triangle, radiance-table row) sequence, so a test can tell which pixels
hold such a "flipped" sample, and the table of the literal (libm) oracle against the reference: THE REFERENCE'S OWN NOISE FLOOR against a
second restatement of its formulas (0 ... 10 flipped samples of 1.23 M per lobe; whole-map rel-L2 6e-8 ... 2.5e-4).

  not gpu: the literal oracle reproduces the table stored in the fixture (flipped pixels identical, rel-L2 without them <= 1e-6)
  gpu:     per map, HIP <-> reference and oracle <-> reference, whole map and without the flipped pixels; asserted:
           (i)   HIP <-> reference without the flipped pixels <= 1e-6 (rounding only), on every map
           (ii)  the HIP path stays at the reference's own noise floor: flipped pixels over all lobes <= 1.25 x the literal oracle's + 4, worst whole-map
                 rel-L2 <= the literal oracle's worst (+ rounding), at least as many maps within north_star's 1e-4 as the literal oracle has.  Round 4: the
                 samplers evaluate sin / cos of the ROUNDED polar angle as the reference does (specified asin / acos, a practically correctly rounded sincos
                 in double) instead of closed forms and a ~1-ulp f32 sincos: measured 41 flipped pixels against 40 for the literal oracle (round 3: 64
                 against 40), the worst map the SAME map with the same value (1.1478e-3: samples on which libm and torch-CPU themselves disagree), 10 of 13
                 maps within 1e-4 for both (profiles/r4_parity_room.json, r4_parity_room_reference.json, r4_flip_attribution.json)
           (iii) HIP == the device-arithmetic oracle bit for bit (maps, triangles, table rows)
           The table goes to gpurun_out/parity_room.json (kept as profiles/r4_parity_room.json)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import REPO, golden

sys.path.insert(0, os.path.join(REPO, "tools"))


def _fixture():
    import golden_codec as mg
    g = golden("bake_room.npz")
    room, slf_np, emi_np, K, c2w = mg.workload(mg.ROOM)
    assert np.array_equal(K, g["K"]) and np.array_equal(c2w, g["c2w"])
    return mg, g, room, slf_np, emi_np, K, c2w


def _names(lobe):
    return ["Ld"] if lobe == 0 else [f"Ls0_r{lobe - 1}", f"Ls1_r{lobe - 1}"]


def _oracle_rows(oracle_mod, mg, g, room, slf_np, emi_np, pos, nrm, wo, pix):
    """literal oracle against the reference maps of the fixture -> rows + the oracle's flipped-pixel masks per lobe"""
    osc = oracle_mod.Scene(room["vertices"], room["faces"])
    oslf = oracle_mod.VoxelSLF(slf_np["inds"], slf_np["radiance"], slf_np["voxel_min"], slf_np["voxel_max"])
    oem = oracle_mod.SLFEmitter(emi_np["is_emitter"], emi_np["emitter_radiance"], emi_np["emitter_area"], oslf)
    P, spp = int(g["P"]), int(g["spp"])
    rows, masks, handles = [], [], (osc, oem)
    for lobe in range(7):
        kw = {} if lobe == 0 else {"wo": wo, "roughness": np.float32(g["roughness_level"][lobe - 1])}
        lit = oracle_mod.bake(osc, oem, pos, nrm, spp, seed=int(g["seed"]), stream=lobe, pix_id=pix, want_tri=True, want_src=True, **kw)
        flip_px = mg.sample_hash(lit[-2], lit[-1], P, spp) != g["sample_hash"][lobe]
        masks.append(flip_px)
        for m, name in enumerate(_names(lobe)):
            rows.append({"map": name, "oracle_vs_reference_flipped_pixels": int(flip_px.sum()), "oracle_vs_reference_rel_l2": mg.rel(lit[m], g[name]),
                         "oracle_vs_reference_rel_l2_without_flipped_pixels": mg.rel(lit[m], g[name], ~flip_px)})
    return rows, masks, handles


def test_literal_oracle_reproduces_the_reference_table(oracle_mod):
    mg, g, room, slf_np, emi_np, K, c2w = _fixture()
    osc = oracle_mod.Scene(room["vertices"], room["faces"])
    H, W = int(g["H"]), int(g["W"])
    xs, ds = oracle_mod.raygen_real(K, c2w, H, W)
    pos, nrm, _, idx, valid = osc.ray_intersect(xs, ds)
    pix = np.nonzero(valid)[0].astype(np.int32)
    assert np.array_equal(pix, g["pix_id"])
    # the oracle's own primary pass agrees with the reference's tensors to rounding (ray generation is pinned to <= 2e-6, not bit for bit) ...
    np.testing.assert_allclose(pos[valid], g["position"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(nrm[valid], g["normal"], rtol=0, atol=1e-5)
    # ... and the bake starts from the reference's tensors, so that only the loop bodies are compared
    rows, _, _ = _oracle_rows(oracle_mod, mg, g, room, slf_np, emi_np, g["position"], g["normal"], g["wo"], pix)
    stored = {r["map"]: r for r in json.loads(str(g["oracle_vs_reference"]))}
    for r in rows:
        s = stored[r["map"]]
        assert r["oracle_vs_reference_flipped_pixels"] == s["oracle_vs_reference_flipped_pixels"], (r, s)
        assert r["oracle_vs_reference_rel_l2_without_flipped_pixels"] <= 1e-6
        assert abs(r["oracle_vs_reference_rel_l2"] - s["oracle_vs_reference_rel_l2"]) <= 1e-9 + 1e-6 * s["oracle_vs_reference_rel_l2"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_hip_vs_reference_python_room_scale(oracle_mod):
    import argparse
    import torch
    import bench
    from iris_amd import bake_shading as bs
    from iris_amd.utils.dataset import real_ldr
    mg, g, room, slf_np, emi_np, K, c2w = _fixture()
    dev = torch.device("cuda:0")
    args = argparse.Namespace(scene_seed=int(g["seed"]), tris=int(g["tris"]), slf_res=int(g["slf_res"]), layout=0, long_walls=False)
    room2, _, _, scene, emitter = bench.build_workload(args, dev)
    assert np.array_equal(room2["faces"], room["faces"])
    H, W, spp, P = int(g["H"]), int(g["W"]), int(g["spp"]), int(g["P"])
    xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
    gb = bs.primary_hits(scene, xs, ds)
    pix = gb["pix_id"].cpu().numpy().astype(np.int32)
    assert np.array_equal(pix, g["pix_id"]), "primary hits differ from the reference replay"
    np.testing.assert_allclose(gb["position"].cpu().numpy(), g["position"], rtol=0, atol=2e-5)    # HIP primary pass vs the reference's tensors: rounding
    np.testing.assert_allclose(gb["normal"].cpu().numpy(), g["normal"], rtol=0, atol=1e-5)
    pos, nrm, wo = g["position"], g["normal"], g["wo"]                                             # the bake starts from the reference's tensors
    gb = {"position": torch.from_numpy(pos).to(dev), "normal": torch.from_numpy(nrm).to(dev), "wo": torch.from_numpy(wo).to(dev), "pix_id": gb["pix_id"]}
    orows, omasks, (osc, oem) = _oracle_rows(oracle_mod, mg, g, room, slf_np, emi_np, pos, nrm, wo, pix)
    rough = [float(r) for r in g["roughness_level"]]
    table, k = [], 0
    for lobe in range(7):
        if lobe == 0:
            hip = bs.bake_diffuse(scene, emitter, gb["position"], gb["normal"], spp, seed=int(g["seed"]), stream_id=0, pix_id=gb["pix_id"], want_tri=True, want_src=True)
            kw = {}
        else:
            hip = bs.bake_specular(scene, emitter, gb["position"], gb["normal"], gb["wo"], rough[lobe - 1], spp, seed=int(g["seed"]), stream_id=lobe,
                                   pix_id=gb["pix_id"], want_tri=True, want_src=True)
            kw = {"wo": wo, "roughness": np.float32(rough[lobe - 1])}
        hip = [t.cpu().numpy() for t in hip]
        with oracle_mod.device_arithmetic():                                         # (iii)
            dev_o = oracle_mod.bake(osc, oem, pos, nrm, spp, seed=int(g["seed"]), stream=lobe, pix_id=pix, want_tri=True, want_src=True, **kw)
        for a, b in zip(hip, dev_o):
            np.testing.assert_array_equal(a, b)
        flip_px = mg.sample_hash(hip[-2], hip[-1], P, spp) != g["sample_hash"][lobe]
        for m, name in enumerate(_names(lobe)):
            row = dict(orows[k]); k += 1
            row.update({"hip_vs_reference_flipped_pixels": int(flip_px.sum()), "hip_vs_reference_rel_l2": mg.rel(hip[m], g[name]),
                        "hip_vs_reference_rel_l2_without_flipped_pixels": mg.rel(hip[m], g[name], ~flip_px),
                        "pixels_flipped_by_both": int((flip_px & omasks[lobe]).sum()),
                        "north_star_1e-4_met_by_hip": bool(mg.rel(hip[m], g[name]) <= 1e-4),
                        "north_star_1e-4_met_by_literal_oracle": bool(row["oracle_vs_reference_rel_l2"] <= 1e-4)})
            table.append(row)
            print(row)
    hip_px = sum(r["hip_vs_reference_flipped_pixels"] for r in table if r["map"] == "Ld" or r["map"].startswith("Ls0"))
    orc_px = sum(r["oracle_vs_reference_flipped_pixels"] for r in table if r["map"] == "Ld" or r["map"].startswith("Ls0"))
    worst_hip = max(r["hip_vs_reference_rel_l2"] for r in table)
    worst_orc = max(r["oracle_vs_reference_rel_l2"] for r in table)
    out = {"config": f"tests/golden/bake_room.npz: synth.room({int(g['seed'])}, {int(g['tris'])}) = {room['faces'].shape[0]} triangles, SLF H={int(g['slf_res'])}, "
                     f"view {int(g['view'])}, {W}x{H}, spp {spp} per lobe, {P} valid pixels, {P * spp} samples per lobe; reference = the reference's Python "
                     "(torch-CPU) with the oracle's closest hit; flipped pixel = a pixel whose per-sample (triangle, table row) hash differs from the reference's",
           "summary": {"flipped_pixels_all_lobes": {"hip": hip_px, "literal_oracle": orc_px}, "worst_whole_map_rel_l2": {"hip": worst_hip, "literal_oracle": worst_orc},
                       "maps_within_1e-4": {"hip": sum(r["north_star_1e-4_met_by_hip"] for r in table), "literal_oracle": sum(r["north_star_1e-4_met_by_literal_oracle"] for r in table), "of": len(table)},
                       "worst_rel_l2_without_flipped_pixels": {"hip": max(r["hip_vs_reference_rel_l2_without_flipped_pixels"] for r in table),
                                                               "literal_oracle": max(r["oracle_vs_reference_rel_l2_without_flipped_pixels"] for r in table)}},
           "bars": {"rel_l2_without_flipped_pixels": 1e-6, "flipped_pixels": "hip <= 1.25 x literal oracle + 4", "worst_whole_map_rel_l2": "hip <= literal oracle", "maps_within_1e-4": "hip >= literal oracle"},
           "maps": table}
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "parity_room.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(out["summary"])
    for r in table:
        assert r["hip_vs_reference_rel_l2_without_flipped_pixels"] <= 1e-6, r            # (i)
    assert hip_px <= 1.25 * orc_px + 4, (hip_px, orc_px)                                 # (ii)
    assert worst_hip <= worst_orc * (1 + 1e-5), (worst_hip, worst_orc)
    assert out["summary"]["maps_within_1e-4"]["hip"] >= out["summary"]["maps_within_1e-4"]["literal_oracle"], out["summary"]
