#!/usr/bin/env python3
"""How the BRDF-sampling + closest-hit stage of the path-tracing integrators (iris_pt_brdf_trace: utils/path_tracing.py:384-391) scales with the number of rays in a call,
on the bench scene: the one-ray-per-thread kernel in its phase-scheduled and its latency-mode instantiation (iris_debug_set joint_max_rays) against the direction-sorted
tile kernel (iris_debug_set pt_tile_min).  A launch of N rays is N / 64 waves over 1024 SIMDs:
below ~0.4 M rays (6 waves per SIMD) it is ONE round whose length is its longest wave -- latency, not issue -- which is why cfg 5's 262 144-ray calls run at a tenth of the
bake kernel's ray rate and why one call of spp = SPP is twice as fast per path.  Prints one JSON line."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import bench
    from iris_amd import _lib as L, bake_shading as bs
    from iris_amd.utils.dataset import real_ldr
    from tools import synth
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(scene_seed=1, tris=1_000_000, slf_res=64, layout=0, long_walls=False)
    room, _, _, scene, _ = bench.build_workload(ns, dev)
    H, W = 1080, 1920
    K, c2w = synth.camera(H, W, 0)
    xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
    g = bs.primary_hits(scene, xs, ds, image_width=W, block=8)
    gen = torch.Generator(device="cpu").manual_seed(0)
    rows = []
    lib = L.lib()
    for n_px, spp in ((1024, 32), (2048, 32), (4096, 32), (8192, 32), (16384, 32), (32768, 32), (65536, 32), (131072, 32)):
        pick = torch.randint(0, g["position"].shape[0], (n_px,), generator=gen).to(dev)
        pos = g["position"][pick].repeat_interleave(spp, 0).contiguous(); nrm = g["normal"][pick].repeat_interleave(spp, 0).contiguous(); wo = g["wo"][pick].repeat_interleave(spp, 0).contiguous()
        N = pos.shape[0]
        alb = torch.full((N, 3), 0.5, device=dev); rough = torch.full((N,), 0.5, device=dev); metal = torch.zeros(N, device=dev)
        s1, s2 = torch.rand(N, device=dev), torch.rand(N, 2, device=dev)
        wi = torch.empty(N, 3, device=dev); pdf = torch.empty(N, device=dev); w = torch.empty(N, 3, device=dev); pn = torch.empty(N, 3, device=dev); nn = torch.empty(N, 3, device=dev)
        tri = torch.empty(N, device=dev, dtype=torch.int64); hit = torch.empty(N, device=dev, dtype=torch.bool)
        row = {"rays": N, "waves_per_simd_if_all_resident": round(N / 64 / 1024, 2)}
        ref = None
        for name, tile_min, joint in (("one_ray_per_thread", 1 << 40, 0), ("one_ray_per_thread_latency_mode", 1 << 40, 1 << 40), ("tile_sorted", 0, 0)):
            L.debug_set("pt_tile_min", tile_min); L.debug_set("joint_max_rays", joint)

            def call():
                L.check(lib.iris_pt_brdf_trace(scene.handle, L.ptr(pos), L.ptr(nrm), L.ptr(wo), L.ptr(alb), L.ptr(rough), L.ptr(metal), L.ptr(s1), L.ptr(s2), N,
                                               L.ptr(wi), L.ptr(pdf), L.ptr(w), L.ptr(pn), L.ptr(nn), L.ptr(tri), L.ptr(hit), 0, 0.0, L.stream()))
            call(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                call()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.steps
            row[name] = {"ms": round(ms, 4), "mrays_per_s": round(N / ms / 1e3, 1)}
            if ref is None:
                ref = tri.clone()
            else:
                row["same_hits"] = row.get("same_hits", True) and bool(torch.equal(ref, tri))
        L.debug_set("pt_tile_min", -1); L.debug_set("joint_max_rays", -1)
        rows.append(row)
    print(json.dumps({"stage": "iris_pt_brdf_trace (sample_brdf + closest hit), bench scene 1.0 M triangles, random pixels of view 0 x spp 32", "rows": rows}))


if __name__ == "__main__":
    main()
