#!/usr/bin/env python3
"""Time the denoiser on one 1080p view's 11 maps (diffuse + 5 roughness levels x 2), guides from a real primary pass."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--tris", type=int, default=200_000)
    args = ap.parse_args()
    import bench
    from iris_amd import bake_shading as bs
    from iris_amd.utils.dataset import real_ldr
    from iris_amd.utils.denoise import Denoiser
    from tools import synth
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(scene_seed=1, tris=args.tris, slf_res=128, layout=0)
    room, slf, emi, scene, emitter = bench.build_workload(ns, dev)
    H, W = 1080, 1920
    K, c2w = synth.camera(H, W, 0)
    xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
    out = bs.bake_view(scene, emitter, xs, ds, 16, [16] * 6, image_width=W)
    from iris_amd.utils.path_tracing import ray_intersect
    pos, nrm, _, _, valid = ray_intersect(scene, xs, ds)
    dn = Denoiser((W, H), dev).set_guides(nrm, pos, valid)
    maps = [out["diffuse"]] + [out[k][i] for i in range(1, 6) for k in ("specular0", "specular1")]
    dn.denoise_maps(maps); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dn.denoise_maps(maps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": "denoise 11 maps of one 1080p view", "ms": round(dt * 1e3, 2), "ms_per_map": round(dt * 1e3 / 11, 2),
                      "Mpixels/s": round(11 * H * W / dt / 1e6, 1)}))


if __name__ == "__main__":
    main()
