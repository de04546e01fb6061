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


class TimedMaterial:
    """wraps a material network: HIP events around every evaluation -> the share of a pass spent in it, and the points it was asked for"""
    def __init__(self, net):
        self.net, self.ev, self.points = net, [], 0
        if hasattr(net, "roughness_min"):
            self.roughness_min = net.roughness_min

    def __call__(self, position):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = self.net(position); e1.record()
        self.ev.append((e0, e1)); self.points += int(position.shape[0])
        return out

    def ms(self):
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.ev)


def run_f1(scene, emitter, slf, dev, batch_pixels=10240, batches=8, spp=128, depth=5, material="ngp", pos=None, ds=None, nrm=None, tri=None):
    """refine_shading's diffuse pass (refine_shading.py:99-131) on `batches` batches of `batch_pixels` primary hits: first-bounce paths per second, and how much of
    the pass is the material network (every bounce evaluates it at scattered hit points: the stage the pass is bound by)"""
    from iris_amd.utils.path_tracing import path_tracing_det_diff, ray_intersect
    from iris_amd.utils.dataset import real_ldr
    from tools import synth
    H, W = 1080, 1920
    if pos is None:
        K, c2w = synth.camera(H, W, 0)
        xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
        pos, nrm, _, tri, _ = ray_intersect(scene, xs, ds)
    mat = TimedMaterial(ngp_material(slf, dev) if material == "ngp" else GpuStub())
    bp = batch_pixels

    def f1(n_batches):
        for b in range(n_batches):
            b0 = b * bp * 7 % (H * W - bp)          # spread the batches over the image
            path_tracing_det_diff(scene, emitter, mat, pos[b0:b0 + bp], ds[b0:b0 + bp], nrm[b0:b0 + bp], None, tri[b0:b0 + bp], spp, depth)
    f1(1); torch.cuda.synchronize()
    mat.ev, mat.points = [], 0
    t0 = time.perf_counter(); f1(batches); torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / batches
    ms_mat = mat.ms() / batches
    # where a batch goes: HIP events around the stages of ONE more batch (iris_amd/_lib.py StageTimer; the material network's wrapper events are not recorded here)
    from iris_amd import _lib as L_
    with L_.StageTimer() as tm:
        f1(1)
    stage_ms = {k: round(v, 3) for k, v in sorted(tm.ms().items(), key=lambda kv: -kv[1])}
    return {"stages_ms_of_one_batch": stage_ms, "stages_sum_ms": round(sum(stage_ms.values()), 2),"material": material, "pixels_per_batch": bp, "spp": spp, "indir_depth": depth, "ms_per_batch": round(t * 1e3, 2), "Mpaths_per_s": round(bp * spp / t / 1e6, 1),
            "material_network": {"ms_per_batch": round(ms_mat, 2), "share_of_the_pass": round(ms_mat / (t * 1e3), 3), "points_per_batch": mat.points // batches,
                                 "mpoints_per_s": round(mat.points / batches / ms_mat / 1e3, 1) if ms_mat > 0 else None,
                                 "note": "every bounce evaluates the network at the (scattered) hit points of the paths still alive: 1.6 Gpoints/s is the encoding's rate on scattered positions "
                                         "(bound by L2 requests, DESIGN.md 5f)"},
            "note": "first-bounce paths (each continues up to indir_depth bounces with NEE)"}


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
    f1_row = run_f1(scene, emitter, slf, dev, args.batch_pixels, args.batches, args.spp, args.depth, args.material, pos=pos, ds=ds, nrm=nrm, tri=tri)

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
        "f1_refine_diffuse": f1_row,
        "f2_bake_slf": {"views": args.views, "pixels": args.views * H * W, "seconds": round(t_slf, 3), "Mpixels_per_s": round(args.views * H * W / t_slf / 1e6, 1),
                        "occupied_voxels": int(sd["mask"].sum()), "note": "3 passes over the views (bounds, occupancy, pooling); every view is traced once and its hits are kept"},
        "f2_extract_emitters": {"seconds": round(t_em, 3), "Mpixels_per_s": round(args.views * H * W / t_em / 1e6, 1), "emitters": int(em["is_emitter"].sum())}}))


if __name__ == "__main__":
    main()
