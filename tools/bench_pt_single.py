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


def run(room, slf, emi, scene, emitter0, dev, steps=10, warmup=2, rays=8192, spp=32, calls=4, graph=False, material="ngp"):
    """-> dict: Mpaths/s of `steps` training steps (each `calls` forward calls + one backward) on an existing bench workload.
    material: "ngp" = NGPBRDF with random parameters (the reference's network), "stub" = the closed-form stand-in of rounds 1-3"""
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

    def step():
        em.radiance.grad = None
        loss = 0
        for _ in range(calls):
            L = path_tracing_single(scene, em, mat, o, d, dx, dy, spp)
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
                    L = path_tracing_single(scene, em, mat, o, d, dx, dy, spp)
                    loss = loss + ((L - target) ** 2).mean()
                loss.backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            em.radiance.grad.zero_()
            loss = 0
            for _ in range(calls):
                L = path_tracing_single(scene, em, mat, o, d, dx, dy, spp)
                loss = loss + ((L - target) ** 2).mean()
            loss.backward()
        step = g_.replay
        step(); torch.cuda.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    paths = steps * calls * rays * spp
    return {"metric": "path_tracing_single fwd+bwd (BASELINE configs[4]: train_emitter.py:181-189)", "value": round(paths / dt / 1e6, 2), "unit": "Mpaths/s",
            "ms_per_step": round(dt / steps * 1e3, 2),
            "config": {"rays": rays, "spp": spp, "calls_per_step": calls, "hip_graph": bool(graph), "triangles": int(room["faces"].shape[0]), "material": ("NGPBRDF (hash grid 32 x 2 x 2^19 + MLP 64 x 2 on the matrix cores), random parameters" if material == "ngp" else "closed-form stub")},
            "grad_nonzero_rows": int((em.radiance.grad.abs().sum(-1) > 0).sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=8192); ap.add_argument("--spp", type=int, default=32); ap.add_argument("--calls", type=int, default=4)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--graph", action="store_true", help="capture the training step in a HIP graph (torch.cuda.CUDAGraph) and replay it")
    ap.add_argument("--material", choices=["ngp", "stub"], default="ngp")
    ap.add_argument("--pt-tile-min", type=int, default=-1, help="iris_debug_set pt_tile_min: calls of at least this many rays go through the tiled tracing stages (default: the library's)")
    args = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    if args.pt_tile_min >= 0:
        from iris_amd import _lib as L
        L.debug_set("pt_tile_min", args.pt_tile_min)
    ns = argparse.Namespace(scene_seed=1, tris=args.tris, slf_res=256, layout=0)
    room, slf, emi, scene, emitter0 = bench.build_workload(ns, dev)
    print(json.dumps(run(room, slf, emi, scene, emitter0, dev, args.steps, args.warmup, args.rays, args.spp, args.calls, graph=args.graph, material=args.material)))


if __name__ == "__main__":
    main()
