#!/usr/bin/env python3
"""Parameter sweep of the a-trous denoiser: PSNR of denoised low-spp maps against a high-spp bake (synthetic room)."""
import argparse, itertools, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def psnr(a, b):
    a = a.double(); b = b.double()
    return float(10 * torch.log10(b.max() ** 2 / ((a - b) ** 2).mean().clamp_min(1e-30)))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--spp", type=int, default=16); ap.add_argument("--ref-spp", type=int, default=4096)
    args = ap.parse_args()
    import bench
    from iris_amd import bake_shading as bs
    from iris_amd.utils.dataset import real_ldr
    from iris_amd.utils.denoise import Denoiser
    from iris_amd.utils.path_tracing import ray_intersect
    from tools import synth
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(scene_seed=0, tris=200_000, slf_res=128, layout=0)
    room, slf, emi, scene, emitter = bench.build_workload(ns, dev)
    H, W = 480, 640
    res = {}
    for view in (0, 11):
        K, c2w = synth.camera(H, W, view)
        xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
        ref = bs.bake_view(scene, emitter, xs, ds, args.ref_spp, [args.ref_spp] * 6, seed=101, image_width=W)
        raw = bs.bake_view(scene, emitter, xs, ds, args.spp, [args.spp] * 6, seed=7, image_width=W)
        pos, nrm, _, _, valid = ray_intersect(scene, xs, ds)
        maps = {"diffuse": (raw["diffuse"], ref["diffuse"]), "s0_l2": (raw["specular0"][2], ref["specular0"][2]), "s0_l5": (raw["specular0"][5], ref["specular0"][5]),
                "s1_l3": (raw["specular1"][3], ref["specular1"][3])}
        for it, sl, sn, sp in itertools.product((4, 5, 6), (4.0, 8.0, 16.0, 32.0), (64.0, 128.0, 256.0), (0.02, 0.05, 0.1)):
            dn = Denoiser((W, H), dev, iterations=it, sigma_l=sl, sigma_n=sn, sigma_p=sp).set_guides(nrm, pos, valid)
            outs = dn.denoise_maps([m[0] for m in maps.values()])
            for (name, (r_, rf)), o in zip(maps.items(), outs):
                res.setdefault((it, sl, sn, sp), []).append(psnr(o.reshape(-1, 3), rf))
        res.setdefault("raw", []).extend(psnr(r_, rf) for r_, rf in maps.values())
    raw = res.pop("raw")
    rows = sorted(((float(np.mean(v)), k, [round(x, 2) for x in v]) for k, v in res.items()), reverse=True)
    print("raw mean %.2f" % np.mean(raw), [round(x, 2) for x in raw])
    for m, k, v in rows[:8]:
        print("%.2f" % m, k, v)
    print("default (5,16,128,0.05): %.2f" % np.mean(res[(5, 16.0, 128.0, 0.05)]), [round(x, 2) for x in res[(5, 16.0, 128.0, 0.05)]])
    print("worst %.2f" % rows[-1][0], rows[-1][1])


if __name__ == "__main__":
    main()
