#!/usr/bin/env python3
"""Throughput of the material network NGPBRDF.forward (iris_amd/csrc/iris_ngp.h: hash-grid gathers + the perceptron on the matrix cores) on
one MI355X, random-initialised parameters of the reference's configuration (model/brdf.py:222-241).

    python tools/bench_ngp.py [--points 4194304] [--steps 10]

Two input distributions: positions drawn uniformly in the scene box (incoherent: every level's gathers scatter over its whole table) and the
primary hits of a 1080p view of the bench room in pixel-block order (what refine_shading feeds: neighbouring pixels share cells on the coarse levels).
Prints one JSON line: Mpoints/s, per-kernel times (HIP events), and the gather roofline -- algorithmic bytes (32 levels x 8 corners x 4 B + position +
outputs + the 2 x 128 B of the feature planes) against the HBM peak, and the 64-B line traffic the gathers put on the L2s (8 lines per level and point
when no two corners share a line) against the guide's random-line L2 rate."""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1 << 22)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    from iris_amd.model.brdf import NGPBRDF
    from iris_amd import _lib as L
    dev = torch.device("cuda:0")
    n_params = int(L.lib().iris_ngp_n_params())
    g = torch.Generator().manual_seed(0)
    net = NGPBRDF(-3.0, 3.0)
    net.load_state_dict({"mlp.params": (torch.rand(n_params, generator=g) * 2 - 1) * 0.3})
    N = args.points
    out = {"kernel": "ngp_encode_kernel + ngp_mlp_kernel (v_mfma_f32_32x32x16_f16)", "points": N, "n_params": n_params, "table_MB_half": round((n_params - 9216) * 2 / 1e6, 1)}

    def run(pos, name):
        net(pos[:1024]); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        net(pos)
        ev[0].record()
        for i in range(args.steps):
            net(pos); ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
        med = ms[len(ms) // 2]
        alg = 32 * 8 * 4 + 12 + 20 + 2 * 128
        out[name] = {"ms_per_call_median": round(med, 3), "mpoints_per_s": round(N / med / 1e3, 1), "algorithmic_bytes_per_point": alg,
                     "roofline": {"bound": "hbm", "achieved": round(alg * N / (med * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg * N / (med * 1e-3) / 1e9 / 8000.0, 4),
                                  "note": "the gathers are 4-B reads of random 64-B lines of a 2 MiB-per-level table: what they load on the machine is line traffic on the L2s, "
                                          "l2_line_GBps below, not HBM bytes (the tables, 56 MB in half, live in L2 / Infinity Cache)"},
                     "l2_line_GBps": round(32 * 8 * 64 * N / (med * 1e-3) / 1e9, 1)}
    run((torch.rand(N, 3, generator=g) * 6 - 3).to(dev), "uniform_positions")
    # primary hits of a view of the bench room
    try:
        import argparse as _ap
        import bench
        from iris_amd import bake_shading as bs
        from iris_amd.utils.dataset import real_ldr
        from tools import synth
        a = _ap.Namespace(scene_seed=1, tris=1_000_000, slf_res=64, layout=0, long_walls=False)
        room, _, _, scene, _ = bench.build_workload(a, dev)
        K, c2w = synth.camera(1080, 1920, 0)
        xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (1080, 1920)), c2w, False, device=dev)
        gb = bs.primary_hits(scene, xs, ds, image_width=1920, block=8)
        pos = gb["position"]
        pos = pos.repeat((N + pos.shape[0] - 1) // pos.shape[0], 1)[:N].contiguous()
        run(pos, "primary_hits_of_a_1080p_view")
        # the same surface points in RANDOM order (what a later bounce of the refine integrators feeds), and what sorting them along a Morton curve first would buy
        # (prototype in torch: 10 bits per axis; sort + gather + scatter-back timed separately)
        perm0 = torch.randperm(N, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        pos_r = pos[perm0].contiguous()
        run(pos_r, "surface_points_random_order")

        bits = int(os.environ.get("NGP_SORT_BITS", "10"))          # bits per axis of the sort key (experiments: how coarse may the ordering be?)

        def morton_perm(p):
            q = ((p + 3.0) * (1024.0 / 6.0)).to(torch.int64).clamp_(0, 1023)
            q = (q >> (10 - bits)) << (10 - bits)
            def spread(v):
                v = (v | (v << 16)) & 0x030000FF
                v = (v | (v << 8)) & 0x0300F00F
                v = (v | (v << 4)) & 0x030C30C3
                v = (v | (v << 2)) & 0x09249249
                return v
            code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
            return torch.sort(code.to(torch.int32)).indices
        pm = morton_perm(pos_r)
        run(pos_r[pm].contiguous(), "surface_points_morton_sorted")
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        ref = net(pos_r)
        e0.record()
        for _ in range(args.steps):
            pm = morton_perm(pos_r)
        e1.record()
        for _ in range(args.steps):
            o = net(pos_r[pm])
            back = {k: torch.empty_like(v).index_copy_(0, pm, v) for k, v in o.items()}
        e2.record()
        torch.cuda.synchronize()
        out["surface_points_morton_sorted"]["sort_ms_torch_prototype"] = round(e0.elapsed_time(e1) / args.steps, 3)
        out["surface_points_morton_sorted"]["gather_forward_scatter_ms"] = round(e1.elapsed_time(e2) / args.steps, 3)
        out["surface_points_morton_sorted"]["same_bits_as_unsorted"] = bool(all(torch.equal(back[k], ref[k]) for k in ref))
    except Exception as e:     # noqa
        out["primary_hits_of_a_1080p_view"] = {"skipped": repr(e)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
