#!/usr/bin/env python3
"""Measurements for SURVEY.md 8(f) rows 1 and 2 on the bench scene (1.0 M-triangle room; material = the reference's NGPBRDF with random
parameters, --material stub for the closed-form stand-in of rounds 1-3):
  f1  refine_shading's diffuse pass (refine_shading.py:99-131): path_tracing_det_diff, spp 128, indir_depth 5, the reference's batches
      of 10240 pixels (--batch-pixels; 288 GB allows far bigger ones) -> first-bounce paths per second
  f2  the pre-bake chain (slf_bake.py:69-145, extract_emitter_ldr.py:72-115): bake_slf + extract_emitters over V 1080p views
Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch


from tools.bench_pt_single import GpuStub, ngp_material


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-pixels", type=int, default=10240); ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--spp", type=int, default=128); ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--views", type=int, default=4); ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--material", choices=["ngp", "stub"], default="ngp")
    args = ap.parse_args()
    import bench
    from iris_amd import slf_bake as sb
    from iris_amd.utils.path_tracing import path_tracing_det_diff, ray_intersect
    from iris_amd.utils.dataset import real_ldr
    from tools import synth
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(scene_seed=1, tris=args.tris, slf_res=256, layout=0)
    room, slf, emi, scene, emitter = bench.build_workload(ns, dev)
    emitter = emitter.to(dev) if hasattr(emitter, "to") else emitter
    H, W = 1080, 1920
    K, c2w = synth.camera(H, W, 0)
    xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
    pos, nrm, uv, tri, valid = ray_intersect(scene, xs, ds)
    mat = ngp_material(slf, dev) if args.material == "ngp" else GpuStub()
    bp = args.batch_pixels

    def f1(n_batches):
        for b in range(n_batches):
            b0 = b * bp * 7 % (H * W - bp)          # spread the batches over the image
            path_tracing_det_diff(scene, emitter, mat, pos[b0:b0 + bp], ds[b0:b0 + bp], nrm[b0:b0 + bp], None, tri[b0:b0 + bp], args.spp, args.depth)
    f1(1); torch.cuda.synchronize(); t0 = time.perf_counter(); f1(args.batches); torch.cuda.synchronize()
    t_f1 = (time.perf_counter() - t0) / args.batches

    views = []
    for v in range(args.views):
        Kv, c2wv = synth.camera(H, W, v * 8)
        o, d = real_ldr.to_world(real_ldr.get_direction(Kv, (H, W)), c2wv, False, device=dev)
        p, _, _, idx, ok = ray_intersect(scene, o, d)
        rgb = 0.3 + 0.2 * torch.sin(p * 2.0); rgb[~ok] = 0
        views.append({"rays": torch.cat([o, d], -1), "rgbs": rgb})
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sd = sb.bake_slf(scene, views, res_spatial=256, dataset="scannetpp", device=dev)
    torch.cuda.synchronize(); t_slf = time.perf_counter() - t0
    t0 = time.perf_counter()
    em = sb.extract_emitters(scene, room["vertices"], room["faces"], views, threshold=5.0, device=dev)
    torch.cuda.synchronize(); t_em = time.perf_counter() - t0
    print(json.dumps({
        "f1_refine_diffuse": {"material": args.material, "pixels_per_batch": bp, "spp": args.spp, "indir_depth": args.depth, "ms_per_batch": round(t_f1 * 1e3, 2),
                              "Mpaths_per_s": round(bp * args.spp / t_f1 / 1e6, 1), "note": "first-bounce paths (each continues up to indir_depth bounces with NEE)"},
        "f2_bake_slf": {"views": args.views, "pixels": args.views * H * W, "seconds": round(t_slf, 3), "Mpixels_per_s": round(args.views * H * W / t_slf / 1e6, 1),
                        "occupied_voxels": int(sd["mask"].sum()), "note": "3 passes over the views (bounds, occupancy, pooling); every view is traced once and its hits are kept"},
        "f2_extract_emitters": {"seconds": round(t_em, 3), "Mpixels_per_s": round(args.views * H * W / t_em / 1e6, 1), "emitters": int(em["is_emitter"].sum())}}))


if __name__ == "__main__":
    main()
