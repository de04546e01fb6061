"""Parity at BASELINE.json configs[1] size against the LITERAL oracle: one 640x480 view of the 1.0 M-triangle room, SPP 64,
all 13 maps (bake_shading.py:108-123 diffuse, :168-188 six specular levels x {Ls0, Ls1}).

north_star asks for <= 1e-4 relative L2 against the reference path.  The integrand is discontinuous (an emitter-edge or voxel-edge
crossing changes a sample by O(1)), so a 1-ulp difference in a sampled direction between the device's arithmetic and libm can flip
rare samples, and a single emitter-edge flip at this size is already ~1e-4 of a map.  What is asserted here is therefore what can
be true, per map:
  (1) HIP == oracle in device-arithmetic mode BIT FOR BIT -- maps, per-sample hit triangles and per-sample radiance-table rows
      (19.7 M samples per lobe);
  (2) against the LITERAL oracle (mode 0: libm, the reference's formulas as written): the number of samples whose
      (tri_next, table row) differ -- the flips -- is <= 2.5e-5 of the samples (measured under the watertight contract of round 3:
      0 ... 235 of 19.7 M per lobe = <= 1.2e-5, growing with the lobe's width), and with the flipped pixels excluded the maps agree
      to <= 1e-6 relative L2 (measured <= 1.7e-7: rounding only);
  (3) the whole-map relative L2 is recorded for every map and bounded by 2.5e-3: with the flips in, it is 5e-8 ... 1.2e-4 in the
      round-3 run (4e-7 ... 1.1e-3 in round 2: which emitter edges a handful of flips happen to cross is chance).  What the
      reference's OWN arithmetic does on such samples is measured in tests/test_parity_room.py (torch-CPU against libm: up to 1.1e-3).
The table goes to gpurun_out/parity_cfg2.json (kept under profiles/)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from conftest import REPO, rel_l2

pytestmark = pytest.mark.gpu

H, W, SPP = 480, 640, 64


def _rel(a, b, mask=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if mask is not None:
        a, b = a[mask], b[mask]
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def parity_table(oracle_mod, scene_seed=1, tris=1_000_000, view=0, long_walls=False, flip_bar=2.5e-5):
    """(config string, per-map rows, all bars met) for one view; also used by tools/parity_more.py for further scenes / views.
    long_walls: the room's six walls added once more as 12 large triangles 2 cm inside (split into clipped references by the builder)."""
    import bench
    from iris_amd import bake_shading as bs
    from iris_amd.utils.dataset import real_ldr
    from tools import synth
    dev = torch.device("cuda:0")
    args = argparse.Namespace(scene_seed=scene_seed, tris=tris, slf_res=256, layout=0, long_walls=long_walls)
    room, slf_np, emi_np, scene, emitter = bench.build_workload(args, dev)
    osc = oracle_mod.Scene(room["vertices"], room["faces"])
    oslf = oracle_mod.VoxelSLF(slf_np["inds"], slf_np["radiance"], slf_np["voxel_min"], slf_np["voxel_max"])
    oem = oracle_mod.SLFEmitter(emi_np["is_emitter"], emi_np["emitter_radiance"], emi_np["emitter_area"], oslf)
    K, c2w = synth.camera(H, W, view)
    xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
    g = bs.primary_hits(scene, xs, ds)
    pos, nrm, wo, pix = (g[k].cpu().numpy() for k in ("position", "normal", "wo", "pix_id"))
    P = pos.shape[0]
    assert P > 0.9 * H * W
    rough = bs.roughness_levels().tolist()
    table, ok = [], True
    for lobe in range(7):
        if lobe == 0:
            hip = bs.bake_diffuse(scene, emitter, g["position"], g["normal"], SPP, seed=0, stream_id=0, pix_id=g["pix_id"], want_tri=True, want_src=True)
            kw = {}
        else:
            hip = bs.bake_specular(scene, emitter, g["position"], g["normal"], g["wo"], rough[lobe - 1], SPP, seed=0, stream_id=lobe, pix_id=g["pix_id"],
                                   want_tri=True, want_src=True)
            kw = {"wo": wo, "roughness": np.float32(rough[lobe - 1])}
        hip = [t.cpu().numpy() for t in hip]
        n_maps = len(hip) - 2
        with oracle_mod.device_arithmetic():
            dev_o = oracle_mod.bake(osc, oem, pos, nrm, SPP, seed=0, stream=lobe, pix_id=pix, want_tri=True, want_src=True, **kw)
        for a, b in zip(hip, dev_o):                                   # (1) bit for bit, maps + triangle ids + table rows
            np.testing.assert_array_equal(a, b)
        lit = oracle_mod.bake(osc, oem, pos, nrm, SPP, seed=0, stream=lobe, pix_id=pix, want_tri=True, want_src=True, **kw)
        tri_h, src_h, tri_l, src_l = hip[-2], hip[-1], lit[-2], lit[-1]
        flip = (tri_h != tri_l) | (src_h != src_l)
        emitter_flip = flip & ((src_h <= -2) | (src_l <= -2))
        flip_px = flip.reshape(P, SPP).any(1)
        for m in range(n_maps):
            name = "Ld" if lobe == 0 else f"Ls{m}_r{lobe - 1}"
            row = {"map": name, "roughness": None if lobe == 0 else round(rough[lobe - 1], 3), "samples": int(flip.size), "flipped_samples": int(flip.sum()),
                   "flipped_triangle": int((tri_h != tri_l).sum()), "flipped_emitter_row": int(emitter_flip.sum()), "flipped_pixels": int(flip_px.sum()),
                   "flip_rate": float(flip.mean()), "rel_l2_whole_map": _rel(hip[m], lit[m]), "rel_l2_without_flipped_pixels": _rel(hip[m], lit[m], ~flip_px),
                   "bit_exact_vs_device_arithmetic_oracle": True}
            table.append(row)
            ok &= row["flip_rate"] <= flip_bar and row["rel_l2_without_flipped_pixels"] <= 1e-6 and row["rel_l2_whole_map"] <= 2.5e-3
    info = scene.info()
    cfg = (f"BASELINE configs[1]: {W}x{H}, SPP {SPP}, room seed {scene_seed}, view {view}, {room['faces'].shape[0]} triangles"
           f"{' incl. 12 wall triangles' if long_walls else ''} ({info['n_leaf_records']} leaf records), SLF H=256, Philox seed 0, valid pixels {P}")
    return cfg, table, ok


@pytest.mark.timeout(1500)
def test_cfg2_all_maps_vs_literal_oracle(oracle_mod):
    cfg, table, ok = parity_table(oracle_mod)
    out = {"config": cfg,
           "bars": {"flip_rate": 2.5e-5, "rel_l2_without_flipped_pixels": 1e-6, "rel_l2_whole_map": 2.5e-3, "north_star_rel_l2": 1e-4}, "maps": table}
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "parity_cfg2.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for r in table:
        print(r)
    assert ok, [r for r in table if r["flip_rate"] > 2.5e-5 or r["rel_l2_without_flipped_pixels"] > 1e-6 or r["rel_l2_whole_map"] > 2.5e-3]
