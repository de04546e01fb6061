#!/usr/bin/env python3
"""Soak test: bake the same 1080p view repeatedly (all 7 lobes, bench scene) and compare every output bit with the first run; the
persistent kernels are dynamically scheduled (tile queue, lane refill), so any ordering bug would show up as a flipped bit."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--runs", type=int, default=40); ap.add_argument("--tris", type=int, default=1_000_000)
    args = ap.parse_args()
    import bench
    from iris_amd import bake_shading as bs
    from iris_amd.utils.dataset import real_ldr
    from tools import synth
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(scene_seed=1, tris=args.tris, slf_res=256, layout=0)
    room, slf, emi, scene, emitter = bench.build_workload(ns, dev)
    H, W = 1080, 1920
    levels = bs.roughness_levels().tolist()
    bad = 0
    for view in (0, 13):
        K, c2w = synth.camera(H, W, view)
        xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
        g = bs.primary_hits(scene, xs, ds, image_width=W)
        ref = None
        for it in range(args.runs):
            res = bs.bake_lobes(scene, emitter, g["position"], g["normal"], g["wo"], [None] + levels, [128] * 7, seed=3, pix_id=g["pix_id"])
            flat = torch.cat([res[0].reshape(-1)] + [t.reshape(-1) for r in res[1:] for t in r])
            if ref is None:
                ref = flat.clone()
                per = bs.bake_specular(scene, emitter, g["position"], g["normal"], g["wo"], levels[3], 128, seed=3, stream_id=4, pix_id=g["pix_id"])
                ok = torch.equal(per[0], res[4][0]) and torch.equal(per[1], res[4][1])
                print("view", view, "per-lobe kernel == view kernel:", ok); bad += 0 if ok else 1
            elif not torch.equal(flat, ref):
                bad += 1
                print("MISMATCH view", view, "run", it, int((flat != ref).sum()), "values differ")
        print("view", view, args.runs, "runs done")
    print("SOAK", "FAILED" if bad else "OK")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
