#!/usr/bin/env python3
"""Round 6 experiment: what does a WINDOW sort of the rays buy on the machine, in the traversal alone?  (EXPERIMENTS.md round 6; the decision itself was taken in tools/bvh_eval/wavesim.)

The tile machinery of the bake (iris_tile.h: per-tile direction sort + persistent-lane traversal) is also behind `iris_pt_brdf_trace`; with lobe 1 it samples the
bake's diffuse rays (cosine lobe around the pixel's normal, origin = the pixel's position).  So the same kernel can be fed the bake's rays in two orders:

  A  as the bake cuts them: 8 x 8-pixel blocks, pixel-major, 4096 consecutive rays (32 pixels x spp 128) per tile
  B  sorted over WINDOWS of 16 x 16 (or 32 x 32) pixels by a fine direction key (octant | NU x NV cells) before the launch: a 4096-ray tile is then a contiguous piece
     of a window's sorted list, and the kernel's own 256-bin sort inside the tile keeps the fine order inside a bin (arrival order)

and timed with HIP events.  The permutation itself is NOT timed: the question is the upper bound of what reordering can return in the traversal phase on real
hardware (caches, clocks and all), next to the simulator's instruction counts.  Prints one JSON line.
    python tools/exp_window_sort.py [--frac 8] [--window 16] [--nu 32 --nv 16]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def dir_key(wi, nu, nv):
    ax, ay, az = wi[:, 0].abs(), wi[:, 1].abs(), wi[:, 2].abs()
    inv = 1.0 / (ax + ay + az + 1e-30)
    a, b = ax * inv, ay * inv
    v = b / (1.0 - a + 1e-30)
    iu = (a * nu).long().clamp_(max=nu - 1); iv0 = (v * nv).long().clamp_(max=nv - 1)
    iv = torch.where(iu % 2 == 1, nv - 1 - iv0, iv0)
    octant = (wi[:, 0] < 0).long() | ((wi[:, 1] < 0).long() << 1) | ((wi[:, 2] < 0).long() << 2)
    return (octant * nu + iu) * nv + iv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frac", type=int, default=8, help="use every frac-th 64 x 64-pixel block of the view")
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--windows", type=str, default="8x4:8:4,16:32:16,32:64:32", help="comma list of WINDOW:NU:NV (WINDOW = side in pixels, or 8x4 = the shipped tile)")
    ap.add_argument("--repeats", type=int, default=5)
    args = ap.parse_args()
    import argparse as _ap
    import bench
    from iris_amd import _lib as L
    from iris_amd import bake_shading as bs
    from iris_amd.model.brdf import BaseBRDF
    from iris_amd.utils.dataset import real_ldr
    from iris_amd.utils.path_tracing import _lobe_trace
    from tools import synth
    dev = torch.device("cuda:0")
    ns = _ap.Namespace(scene_seed=1, tris=1_000_000, slf_res=256, layout=0)
    room, slf, emi, scene, emitter = bench.build_workload(ns, dev)
    H, W, spp = 1080, 1920, args.spp
    K, c2w = synth.camera(H, W, 0)
    xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
    # pixels of every frac-th 64 x 64 block, in 8 x 8-block order (what bake_view hands the kernel)
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    blk = (yy // 64) * ((W + 63) // 64) + xx // 64
    keep = (blk % args.frac == 0) & (yy < H // 64 * 64) & (xx < W // 64 * 64)
    pix = (yy * W + xx)[keep]
    g = bs.primary_hits(scene, xs[pix], ds[pix], pixel_ids=pix, image_width=W, block=8)
    P = g["position"].shape[0]
    pid = g["pix_id"].long()
    py, px = pid // W, pid % W
    pos = g["position"].repeat_interleave(spp, 0).contiguous(); nrm = g["normal"].repeat_interleave(spp, 0).contiguous(); wo = g["wo"].repeat_interleave(spp, 0).contiguous()
    N = P * spp
    torch.manual_seed(0)
    s2 = torch.rand(N, 2, device=dev)
    wi_all, _, _ = BaseBRDF().sample_diffuse(s2, nrm)                  # the direction the stage will sample (same kernel arithmetic)

    def run(order):
        a = [t[order].contiguous() if order is not None else t for t in (pos, nrm, wo, s2)]
        ms = []
        for _ in range(args.repeats + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = _lobe_trace(scene, a[0], a[1], a[2], None, None, a[3], 1, 0.0)
            e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        return float(np.median(ms[1:])), out

    rows = []
    base_ms, base_out = run(None)
    rows.append({"order": "as the bake cuts them (8 x 8-pixel blocks, 32 pixels x spp per tile)", "ms": round(base_ms, 3), "mrays_per_s": round(N / base_ms / 1e3, 1)})
    ray_px, ray_py = px.repeat_interleave(spp), py.repeat_interleave(spp)
    for spec in args.windows.split(","):
        wspec, nu, nv = spec.split(":"); nu, nv = int(nu), int(nv)
        if "x" in wspec:
            wx, wy = (int(v) for v in wspec.split("x"))
        else:
            wx = wy = int(wspec)
        win = (ray_py // wy) * ((W + wx - 1) // wx) + ray_px // wx
        key = win * (8 * nu * nv) + dir_key(wi_all, nu, nv)
        order = torch.sort(key, stable=True)[1]
        ms, out = run(order)
        inv = torch.empty_like(order); inv[order] = torch.arange(N, device=dev)
        same = bool(torch.equal(out[5][inv], base_out[5]))              # the same closest hits, whatever the order
        rows.append({"order": f"sorted over windows of {wx} x {wy} pixels by octant | {nu} x {nv} cells ({8 * nu * nv} bins, {wx * wy * spp} rays per window)", "ms": round(ms, 3),
                     "mrays_per_s": round(N / ms / 1e3, 1), "vs_bake_order": round(base_ms / ms, 4), "same_hits": same})
    print(json.dumps({"what": "iris_pt_brdf_trace (lobe 1 = the bake's diffuse rays) through the tile kernel, the same rays in different orders; the permutation is not timed", "rays": N, "pixels": P, "spp": spp, "rows": rows}))


if __name__ == "__main__":
    main()
