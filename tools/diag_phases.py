#!/usr/bin/env python3
"""Where a tile's time goes: per-phase shader cycles of the tile kernels (diagnostic build -DIRIS_PHASE_TIMING; IRIS_HIP_LIB must point
to it).  Bakes a few views of the bench workload and prints the share of workgroup time per phase:
A sample + bin | B prefix + scatter | C traversal (wave 0) | C' wave 0 waiting for the tile's slowest wave | D shade + reduce."""
import argparse, ctypes as C, json, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--lobes", type=str, default="0,1,2,3,4,5,6")
    args = ap.parse_args()
    import bench
    from iris_amd import _lib as L, bake_shading as bs
    from iris_amd.utils.dataset import real_ldr
    from tools import synth
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(scene_seed=1, tris=1_000_000, slf_res=256, layout=0)
    room, slf, emi, scene, emitter = bench.build_workload(ns, dev)
    lib = C.CDLL(L.LIB_PATH)
    fn = lib.iris_debug_phase_cycles
    fn.argtypes = [C.c_void_p, C.c_int]
    H, W, spp = 1080, 1920, 128
    lobes = [int(x) for x in args.lobes.split(",")]
    rough = bs.roughness_levels().tolist()
    K, _ = synth.camera(H, W, 0)
    out = (C.c_ulonglong * 8)()
    for v in range(args.views + 1):
        c2w = synth.camera(H, W, v * 7, n_views=32)[1]
        xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
        g = bs.primary_hits(scene, xs, ds, image_width=W, block=8)
        if v == 0:
            fn(None, 1)        # warm-up view: reset afterwards
        bs.bake_lobes(scene, emitter, g["position"], g["normal"], g["wo"], [None if l == 0 else rough[l - 1] for l in lobes], [spp] * len(lobes), seed=0, stream_ids=lobes, pix_id=g["pix_id"])
        if v == 0:
            fn(None, 1)
    fn(out, 0)
    c = [int(x) for x in out]
    tot = float(sum(c[:5]))
    names = ["A sample+bin", "B prefix+scatter", "C' wait for slowest wave", "D shade+reduce", "C traversal (wave 0)"]
    print(json.dumps({"cycles": dict(zip(names, c[:5])), "share": {n: round(x / tot, 4) for n, x in zip(names, c[:5])}}))


if __name__ == "__main__":
    main()
