#!/usr/bin/env python3
"""BASELINE cfg 5: path_tracing_single forward + backward throughput (8192 rays x spp 32, 4 calls per step as the reference's
training_step does with SPP=128, train_emitter.py:181-189) on the synthetic 1 M-triangle room, through the reference's material network
(NGPBRDF, random parameters; --material stub: the closed-form stand-in of rounds 1-3).
Prints one JSON line (paths/s = camera paths traced, shaded and back-propagated per second)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch


class GpuStub(torch.nn.Module):
    """closed-form stand-in for NGPBRDF, evaluated on the GPU"""
    def forward(self, x):
        if getattr(self, "_k", None) is None or self._k.device != x.device:        # (constants uploaded once: a host-to-device copy cannot be captured in a graph)
            self._k = torch.tensor([1.3, 2.1, 0.7], device=x.device); self._ph = torch.tensor([0.1, 0.5, 0.9], device=x.device)
        k, ph = self._k, self._ph
        return {"albedo": 0.5 + 0.4 * torch.sin(x * k + ph), "roughness": 0.35 + 0.3 * torch.sin(x[:, :1] * 1.7 + x[:, 1:2] * 0.9),
                "metallic": 0.5 + 0.5 * torch.sin(x[:, 2:3] * 2.3)}


def ngp_material(slf, dev, seed=0):
    """the reference's material network (NGPBRDF, model/brdf.py:213-260: hash grid + MLP, HIP kernels) with RANDOM parameters of its configuration -- there
    is no checkpoint on this machine (no network, no dataset): the architecture, the table sizes and the gather pattern are the real ones"""
    from iris_amd.model.brdf import NGPBRDF
    net = NGPBRDF(slf["voxel_min"], slf["voxel_max"])
    g = torch.Generator().manual_seed(seed)
    net.load_state_dict({"mlp.params": (torch.rand(net.mlp.params.numel(), generator=g) * 2 - 1) * 0.3})
    return net


L2_PEAK_GBPS = 34500.0        # aggregate L2 bandwidth of the 8 XCDs (MI355X_MICROARCH.md: 4 MiB per XCD, about 34.5 TB/s)


def run(room, slf, emi, scene, emitter0, dev, steps=10, warmup=2, rays=8192, spp=32, calls=4, graph=False, material="ngp", skip_unused_material=True, stages=True,
        bake_mrays_per_s=None, streams=1):
    """-> dict: Mpaths/s of `steps` training steps (each `calls` forward calls + one backward) on an existing bench workload.
    material: "ngp" = NGPBRDF with random parameters (the reference's network), "stub" = the closed-form stand-in of rounds 1-3
    skip_unused_material: path_tracing_single's default (True): the network's second evaluation per call, whose only use is a test its roughness bound decides,
    is not launched (same outputs); False = evaluated as the reference does
    stages: also time the stages of a call with HIP events (one extra instrumented step) and price the two leaders: the material network against the L2 line
    roof, the BRDF-sampled rays against the bake kernel's ray rate on the same scene (bake_mrays_per_s)"""
    from iris_amd.model.emitter import SLFEmitterLearn
    from iris_amd.utils.path_tracing import path_tracing_single
    from iris_amd.utils.dataset import real_ldr
    from tools import synth
    import tempfile
    tmp = tempfile.mkdtemp()
    ep, sp = os.path.join(tmp, "emitter.pth"), os.path.join(tmp, "vslf.npz")
    torch.save({"is_emitter": torch.from_numpy(emi["is_emitter"]), "emitter_vertices": torch.from_numpy(emi["emitter_vertices"]),
                "emitter_area": torch.from_numpy(emi["emitter_area"]), "emitter_normal": torch.zeros(len(emi["emitter_area"]), 3),
                "emitter_radiance": torch.from_numpy(emi["emitter_radiance"])}, ep)
    torch.save({"mask": torch.from_numpy(slf["mask"]), "voxel_min": slf["voxel_min"], "voxel_max": slf["voxel_max"], "weight": emitter0.slf.state_dict()}, sp)
    em = SLFEmitterLearn(ep, sp).to(dev)
    H, W = 1080, 1920
    K, c2w = synth.camera(H, W, 0)
    o, d, dx, dy = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, True, device=dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    pick = torch.randint(0, H * W, (rays,), generator=g).to(dev)
    o, d, dx, dy = o[pick], d[pick], dx[pick], dy[pick]
    mat = ngp_material(slf, dev) if material == "ngp" else GpuStub()
    target = torch.rand(rays, 3, device=dev)

    pool = [torch.cuda.Stream(device=dev) for _ in range(streams)] if streams > 1 else []

    def step():
        em.radiance.grad = None
        loss = 0
        if pool:
            # the `calls` forward passes of a step are independent of each other (the same rays, fresh draws: train_emitter.py:181-189 runs them one after the other only
            # because it is a Python loop): issued round-robin on `streams` HIP streams, the latency-bound stages of one call run beside those of the next
            main = torch.cuda.current_stream(dev)
            ev = torch.cuda.Event(); ev.record(main)
            Ls = []
            for c in range(calls):
                st = pool[c % streams]
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    Ls.append(path_tracing_single(scene, em, mat, o, d, dx, dy, spp, skip_unused_material=skip_unused_material))
            for st in pool:
                main.wait_stream(st)
            for L in Ls:
                L.record_stream(main)
                loss = loss + ((L - target) ** 2).mean()
        else:
            for _ in range(calls):
                L = path_tracing_single(scene, em, mat, o, d, dx, dy, spp, skip_unused_material=skip_unused_material)
                loss = loss + ((L - target) ** 2).mean()
        loss.backward()
        return loss
    for _ in range(warmup):
        step()
    if graph:
        # the whole training step -- `calls` forward passes, the loss, the backward scatter -- captured once as a HIP graph and replayed: every launch of
        # the path (ctypes or torch) is stream-ordered and allocates only through torch's graph pool; torch.rand draws advance with every replay
        em.radiance.grad = torch.zeros_like(em.radiance)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                em.radiance.grad.zero_()
                loss = 0
                for _ in range(calls):
                    L = path_tracing_single(scene, em, mat, o, d, dx, dy, spp, skip_unused_material=skip_unused_material)
                    loss = loss + ((L - target) ** 2).mean()
                loss.backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            em.radiance.grad.zero_()
            loss = 0
            for _ in range(calls):
                L = path_tracing_single(scene, em, mat, o, d, dx, dy, spp, skip_unused_material=skip_unused_material)
                loss = loss + ((L - target) ** 2).mean()
            loss.backward()
        step = g_.replay
        step(); torch.cuda.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    paths = steps * calls * rays * spp
    stage_info = None
    if stages and not graph:
        from iris_amd import _lib as L_
        with L_.StageTimer() as tm:
            step()
        per_call = {k: round(v / calls, 4) for k, v in tm.ms().items()}
        n = rays * spp
        stage_info = {"ms_per_call": per_call, "note": "HIP events on the stream each stage runs on, one instrumented step; the NEE stage runs on a side stream BESIDE the BRDF stage, so the "
                      "figures do not add up to the call; `material (sampled hits)` is 0 when the second network evaluation is skipped"}
        t_mat = per_call.get("material (primary hits)", 0.0)
        if material == "ngp" and t_mat > 0:
            # Priced with the L2 -> L1 read traffic the counters of the SAME two kernels show per point (profiles/r4_ngp_pmc.json: TCP_TCC_READ_REQ x 64 B per point of a
            # 2^20-point launch): 3.6 KB for the primary hits of a view (neighbouring samples share cells: L1 hit 0.50), 15.2 KB for positions scattered over the box
            # (L1 hit 0.11).  This evaluation is of the first kind.  (A model of one 64-B line per corner gather, 16 KB per point, over-prices it: > the L2 peak.)
            try:
                pm = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r4_ngp_pmc.json")))["kernels"]["ngp_encode_kernel"]
                b_coh = pm["primary_hits_of_a_1080p_view"]["derived"]["l2_to_l1_read_bytes_per_point"]; b_sc = pm["uniform_positions"]["derived"]["l2_to_l1_read_bytes_per_point"]
            except Exception:     # noqa
                b_coh, b_sc = 3605.3, 15186.7
            gbps = b_coh * n / (t_mat * 1e-3) / 1e9
            stage_info["material_network"] = {"points_per_call": n, "mpoints_per_s": round(n / t_mat / 1e3, 1),
                                              "roofline": {"bound": "l2", "achieved": round(gbps, 1), "peak": L2_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / L2_PEAK_GBPS, 4),
                                                           "l2_to_l1_read_bytes_per_point": {"primary_hits (this evaluation)": b_coh, "scattered_positions": b_sc},
                                                           "frac_if_the_points_were_scattered": round(b_sc * n / (t_mat * 1e-3) / 1e9 / L2_PEAK_GBPS, 4)},
                                              "note": "encode + perceptron of ONE evaluation of 8192 x 32 points at the primary hits; L2 -> L1 read bytes per point from the PMC passes of round 4 on the same "
                                                      "kernels (the encoding's source is unchanged apart from the declared alignment of its pair loads) x this run's point rate, against the aggregate L2 "
                                                      "bandwidth: a launch of 262 144 points (1024 workgroups x 32 levels) is a quarter of the 2^20-point launches the kernel was tuned on"}
        t_tr = per_call.get("brdf sample + trace", 0.0)
        if t_tr > 0:
            stage_info["brdf_rays"] = {"rays_per_call": n, "mrays_per_s": round(n / t_tr / 1e3, 1), "bake_kernel_mrays_per_s_same_scene": bake_mrays_per_s,
                                       "frac_of_bake_kernel_rate": (round(n / t_tr / 1e3 / bake_mrays_per_s, 4) if bake_mrays_per_s else None),
                                       "note": "sampling + one closest hit per path; 262 144 rays are 4 waves per SIMD for ONE round: the launch lasts as long as its longest wave (latency, not "
                                               "issue), which is why one call of spp = SPP (4 x the rays) runs at twice the path rate"}
    return {"metric": "path_tracing_single fwd+bwd (BASELINE configs[4]: train_emitter.py:181-189)", "value": round(paths / dt / 1e6, 2), "unit": "Mpaths/s",
            "ms_per_step": round(dt / steps * 1e3, 2),
            "config": {"rays": rays, "spp": spp, "calls_per_step": calls, "hip_graph": bool(graph), "streams": int(streams), "triangles": int(room["faces"].shape[0]), "material": ("NGPBRDF (hash grid 32 x 2 x 2^19 + MLP 64 x 2 on the matrix cores), random parameters" if material == "ngp" else "closed-form stub")},
            "skip_unused_material": bool(skip_unused_material and material == "ngp"), "stages": stage_info,
            "grad_nonzero_rows": int((em.radiance.grad.abs().sum(-1) > 0).sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=8192); ap.add_argument("--spp", type=int, default=32); ap.add_argument("--calls", type=int, default=4)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--streams", type=int, default=1, help="issue the independent forward calls of a step round-robin on this many HIP streams")
    ap.add_argument("--graph", action="store_true", help="capture the training step in a HIP graph (torch.cuda.CUDAGraph) and replay it")
    ap.add_argument("--material", choices=["ngp", "stub"], default="ngp")
    ap.add_argument("--no-skip", action="store_true", help="evaluate the material network at the sampled hits as the reference does (path_tracing_single skip_unused_material=False)")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE", help="iris_debug_set option (experiments), e.g. joint_max_rays=0")
    ap.add_argument("--pt-tile-min", type=int, default=-1, help="iris_debug_set pt_tile_min: calls of at least this many rays go through the tiled tracing stages (default: the library's)")
    args = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    from iris_amd import _lib as L
    if args.pt_tile_min >= 0:
        L.debug_set("pt_tile_min", args.pt_tile_min)
    for kv in args.debug_set:
        L.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
    ns = argparse.Namespace(scene_seed=1, tris=args.tris, slf_res=256, layout=0)
    room, slf, emi, scene, emitter0 = bench.build_workload(ns, dev)
    print(json.dumps(run(room, slf, emi, scene, emitter0, dev, args.steps, args.warmup, args.rays, args.spp, args.calls, graph=args.graph, material=args.material, skip_unused_material=not args.no_skip, streams=args.streams)))


if __name__ == "__main__":
    main()
