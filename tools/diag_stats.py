#!/usr/bin/env python3
"""Instrumented (`stats`) launches of the bake kernels on the bench workload: prints the raw counters, checks their invariants and
that the instrumented build returns the production build's bits.  IRIS_HIP_LIB selects the library (A/B builds).

    python tools/diag_stats.py [--height 1080 --width 1920 --spp 128 --tris 1000000]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

STAT_NAMES = ["rays", "node_visits", "tri_tests", "wave_node_iters", "wave_leaf_iters", "sp_gt8", "sp_gt12", "sp_gt16", "tail_sum",
              "drain_node_visits", "drain_wave_node_iters", "top21", "top85", "top341", "top1365", "shared_iters", "shared_lanes", "shared_all_iters", "slow_push_iters"]


def check_invariants(st, expected_rays):
    """Invariants of one instrumented launch (st: int64[20]).  Returns a list of violated ones."""
    bad = []
    rays, nodes, tris, nit, lit, g8, g12, g16 = [int(x) for x in st[:8]]
    if rays != expected_rays: bad.append(f"rays {rays} != P*spp {expected_rays}")
    if not (0 <= g16 <= g12 <= g8 <= rays): bad.append(f"stack-depth counters not ordered / in range: {g8} {g12} {g16} rays {rays}")
    if nit * 64 < nodes: bad.append(f"node visits {nodes} > 64 x wave node iterations {nit}")
    if lit * 64 < tris: bad.append(f"triangle tests {tris} > 64 x wave leaf iterations {lit}")
    if nodes < rays: bad.append("fewer node visits than rays (every ray visits the root)")
    if int(st[9]) > nodes or int(st[10]) > nit: bad.append("drain counters exceed the totals")
    t21, t85, t341, t1365 = [int(x) for x in st[11:15]]
    if not (rays <= t21 <= t85 <= t341 <= t1365 <= nodes): bad.append(f"top-of-tree counters not ordered: {rays} {t21} {t85} {t341} {t1365} {nodes}")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--scene-seed", type=int, default=1)
    ap.add_argument("--slf-res", type=int, default=256)
    ap.add_argument("--layout", type=int, default=0)
    ap.add_argument("--lobes", type=str, default="0,1,6")
    args = ap.parse_args()
    import bench
    from iris_amd import bake_shading as bs
    from iris_amd import _lib as L
    from iris_amd.utils.dataset import real_ldr
    from tools import synth
    dev = torch.device("cuda:0")
    room, slf_np, emi_np, scene, emitter = bench.build_workload(args, dev)
    H, W, spp = args.height, args.width, args.spp
    K, c2w = synth.camera(H, W, 0)
    xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
    g = bs.primary_hits(scene, xs, ds, image_width=W, block=8)
    nP = g["position"].shape[0]
    sel = torch.arange(nP, device=dev)
    sel = sel[(sel // 8192) % 16 == 3] if nP > 16 * 8192 else sel
    rough = bs.roughness_levels().tolist()
    out = {"lib": L.LIB_PATH, "pixels": int(sel.numel()), "spp": spp, "launches": []}
    ok = True
    for l in [int(x) for x in args.lobes.split(",")]:
        for variant in (L.BAKE_TILE_SORTED, L.BAKE_PIXEL_PER_WAVE):
            stats = torch.zeros(20, device=dev, dtype=torch.int64)
            if l == 0:
                a = bs.bake_diffuse(scene, emitter, g["position"][sel], g["normal"][sel], spp, seed=0, stream_id=0, pix_id=g["pix_id"][sel], stats=stats, variant=variant)
                b = bs.bake_diffuse(scene, emitter, g["position"][sel], g["normal"][sel], spp, seed=0, stream_id=0, pix_id=g["pix_id"][sel], variant=variant)
                same = bool(torch.equal(a, b))
            else:
                a = bs.bake_specular(scene, emitter, g["position"][sel], g["normal"][sel], g["wo"][sel], rough[l - 1], spp, seed=0, stream_id=l, pix_id=g["pix_id"][sel], stats=stats, variant=variant)
                b = bs.bake_specular(scene, emitter, g["position"][sel], g["normal"][sel], g["wo"][sel], rough[l - 1], spp, seed=0, stream_id=l, pix_id=g["pix_id"][sel], variant=variant)
                same = bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
            torch.cuda.synchronize()
            st = stats.cpu().numpy()
            bad = check_invariants(st, int(sel.numel()) * spp)
            if not same: bad.append("instrumented outputs differ from the production build")
            ok = ok and not bad
            rec = {"lobe": l, "variant": variant, "stats": {n: int(v) for n, v in zip(STAT_NAMES, st)}, "violations": bad}
            r = float(st[0])
            rec["per_ray"] = {"nodes": round(st[1] / r, 3), "tris": round(st[2] / r, 3), "top21": round(st[11] / r, 3), "top85": round(st[12] / r, 3),
                              "top341": round(st[13] / r, 3), "top1365": round(st[14] / r, 3), "lane_util_nodes": round(st[1] / max(st[3] * 64, 1), 3),
                              "sp_gt8": round(st[5] / r, 5), "sp_gt12": round(st[6] / r, 5), "sp_gt16": round(st[7] / r, 6)}
            out["launches"].append(rec)
            print(json.dumps(rec), flush=True)
    print("STATS", "OK" if ok else "VIOLATIONS")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
