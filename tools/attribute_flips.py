#!/usr/bin/env python3
"""Where do the device-arithmetic oracle's (= the HIP kernels') extra disagreements with the reference come from?  Runs the oracle in mode 1 on the
room-scale fixture (tests/golden/bake_room.npz) with single substitutions undone (orc_set_undo) and counts flipped pixels per lobe against the
reference's per-pixel sample hashes.  CPU only.    python tools/attribute_flips.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import oracle                      # noqa: E402
import golden_codec as mg      # noqa: E402


def main():
    g = np.load(os.path.join(REPO, "tests", "golden", "bake_room.npz"))
    room, slf_np, emi_np, K, c2w = mg.workload(mg.ROOM)
    osc = oracle.Scene(room["vertices"], room["faces"])
    oslf = oracle.VoxelSLF(slf_np["inds"], slf_np["radiance"], slf_np["voxel_min"], slf_np["voxel_max"])
    oem = oracle.SLFEmitter(emi_np["is_emitter"], emi_np["emitter_radiance"], emi_np["emitter_area"], oslf)
    P, spp = int(g["P"]), int(g["spp"])
    cases = [("literal (mode 0)", 0, 0), ("device arithmetic (mode 1)", 1, 0), ("mode 1, libm asin/acos", 1, 1), ("mode 1, libm sincos(theta)", 1, 2),
             ("mode 1, libm sincos(phi)", 1, 4), ("mode 1, libm sincos(theta, phi)", 1, 6), ("mode 1, powf", 1, 8), ("mode 1, all libm", 1, 15)]
    out = {}
    for name, mode, undo in cases:
        oracle.set_mode(mode); oracle.lib().orc_set_undo(C.c_int(undo))
        per = []
        for lobe in range(7):
            kw = {} if lobe == 0 else {"wo": g["wo"], "roughness": np.float32(g["roughness_level"][lobe - 1])}
            r = oracle.bake(osc, oem, g["position"], g["normal"], spp, seed=int(g["seed"]), stream=lobe, pix_id=g["pix_id"], want_tri=True, want_src=True, **kw)
            per.append(int((mg.sample_hash(r[-2], r[-1], P, spp) != g["sample_hash"][lobe]).sum()))
        out[name] = {"flipped_pixels_by_lobe": per, "total": sum(per)}
        print(f"{name:36s} {per} total {sum(per)}")
    oracle.set_mode(0); oracle.lib().orc_set_undo(C.c_int(0))
    with open(os.path.join(REPO, "profiles", "r4_flip_attribution.json"), "w") as fh:
        json.dump({"fixture": "tests/golden/bake_room.npz (134 400 pixel-lobes, 8.6 M samples)", "cases": out}, fh, indent=1)


if __name__ == "__main__":
    main()
