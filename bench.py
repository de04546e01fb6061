#!/usr/bin/env python3
"""bench.py -- bake_shading throughput on MI355X (BASELINE.json metric: Mrays/s = shading samples / s).

A "step" is one full bake of one 1920x1080 view of the synthetic ScanNet++-like room (SURVEY.md section 8(d) cfg 3/4):
primary pass + diffuse lobe + 6 specular roughness levels, every lobe at SPP=128, all inputs resident in HBM, uniforms
from the in-kernel Philox stream, plus (N>1) the single all_gather of the 13 maps.  One ray = one (pixel, sample, lobe)
secondary ray traced AND shaded.  N>1 shards the pixels of the SAME view over the ranks (strong scaling).

    python bench.py --gpus N --steps K --warmup W        (N > 1: starts one worker process per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Host synchronisation: the primary pass of a view (`bake_shading.primary_hits`) compacts the valid pixels with a boolean mask, which synchronises the
host ONCE per view (the pixel count is needed to size the launch); everything else of a step is stream-ordered.  It is inside the timed region.

After the timed region rank 0 CHECKS the numbers it timed: a pixel subsample of the LAST timed view's maps (all lobes) against the CPU oracle in
device-arithmetic mode, bit for bit (`parity_check`; a mismatch makes the run fail with exit code 3).

Rank 0 prints ONE JSON line.  `roofline` prices the timed kernel (bake_view_kernel: all lobes of a view behind one launch) against three
calibrated roofs -- VALU issue, the vector-memory (L1 / TA) path, HBM-side bytes -- and names the binding one.
`cpu_baseline` is the CPU oracle (a port of the same algorithm, oracle/) timed on a bounded pixel sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PMC_FILES = ["pmc_r6.json", "pmc_r6_world2.json", "pmc_r6_world4.json", "pmc_r6_world8.json"]     # N = 1, and rank 0's stripes of an N-rank run (EMULATED on one GPU: bench.py --emulate-world N)
N_SIMD, N_CU = 1024, 256


def build_workload(args, dev):
    from tools import synth
    from iris_amd.model.emitter import SLFEmitter
    from iris_amd.utils.path_tracing import Scene
    import tempfile
    room = synth.room(args.scene_seed, args.tris)
    if getattr(args, "long_walls", False):
        # (parity runs) the six walls once more as 12 large triangles 2 cm inside: what a decimated scan looks like to the BVH builder
        v, f = room["vertices"], room["faces"]
        lo, hi = v.min(0) + 0.02, v.max(0) - 0.02
        c = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], v.dtype)
        quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
        nf = np.array([[len(v) + a for a in (q[0], q[1], q[2])] for q in quads] + [[len(v) + a for a in (q[0], q[2], q[3])] for q in quads], f.dtype)
        room = dict(room, vertices=np.concatenate([v, c]), faces=np.concatenate([f, nf]), is_emitter=np.concatenate([room["is_emitter"], np.zeros(len(nf), bool)]))
    slf = synth.slf_for(room["vertices"], room["faces"], args.slf_res)
    emi = synth.emitters_for(room["vertices"], room["faces"], room["is_emitter"])
    tmp = tempfile.mkdtemp(prefix="iris_bench_")
    ep, sp = os.path.join(tmp, "emitter.pth"), os.path.join(tmp, "vslf.npz")
    from iris_amd.model.slf import VoxelSLF
    v = VoxelSLF(torch.from_numpy(slf["mask"]), slf["voxel_min"], slf["voxel_max"])
    v.radiance[:] = torch.from_numpy(slf["radiance"])
    torch.save({"is_emitter": torch.from_numpy(emi["is_emitter"]), "emitter_vertices": torch.from_numpy(emi["emitter_vertices"]),
                "emitter_area": torch.from_numpy(emi["emitter_area"]), "emitter_normal": torch.zeros(len(emi["emitter_area"]), 3),
                "emitter_radiance": torch.from_numpy(emi["emitter_radiance"])}, ep)
    torch.save({"mask": torch.from_numpy(slf["mask"]), "voxel_min": slf["voxel_min"], "voxel_max": slf["voxel_max"], "weight": v.state_dict()}, sp)
    emitter = SLFEmitter(ep, sp)          # the reference's own file formats
    scene = Scene(room["vertices"], room["faces"], device=dev, layout=args.layout)
    emitter.handle(dev); emitter.slf.handle(dev)
    return room, slf, emi, scene, emitter


def spawn_workers(n):
    """`python bench.py --gpus N` as a plain command: start one fresh worker process per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in its environment, the same argv) BEFORE this process makes any GPU call -- a process that has initialised the GPU is
    never re-executed -- and exit with the worst worker's code.  Rank 0's JSON line goes to this process's stdout."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    while procs:
        for p in list(procs):
            c = p.poll()
            if c is None:
                continue
            procs.remove(p)
            if c != 0:
                rc = rc or c
                for q in procs:          # a worker failed: the others would wait in a collective for ever
                    q.terminate()
        time.sleep(0.2)
    sys.exit(rc)


def load_pmc(rays_per_launch, node_bytes):
    """Per-launch counters of the timed kernel from the committed rocprofv3 passes (PMC counters cannot be read from inside this process): the profile among
    PMC_FILES that was taken on exactly the kernel sources this library was built from AND on this launch size -- the whole view at N = 1, rank 0's stripes of an
    N-rank run otherwise (those passes are taken with --emulate-world N on one GPU and say so).  None + the reason when there is no such profile."""
    from iris_amd import _lib as L
    have = L.source_hash()
    why = []
    for name in PMC_FILES:
        try:
            pj = json.load(open(os.path.join(REPO, "profiles", name)))
        except Exception as e:     # noqa
            why.append(f"profiles/{name} unreadable ({type(e).__name__})")
            continue
        if pj.get("source_hash") != have:
            why.append(f"profiles/{name}: stale (taken on kernel sources {pj.get('source_hash')}, this build is {have})")
            continue
        if abs(pj.get("rays_per_launch", 0) - rays_per_launch) > 0.01 * rays_per_launch or node_bytes != 64:
            why.append(f"profiles/{name}: {pj.get('rays_per_launch')} rays per launch, this run launches {rays_per_launch:.0f}")
            continue
        pj["_file"] = name
        return pj, "committed (profiles/" + name + (", taken with --emulate-world %d on one GPU" % pj["emulated_world"] if pj.get("emulated_world") else "") + ")"
    return None, "; ".join(why)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--views", type=int, default=32, help="cameras on the circle (cfg 3); the K timed steps bake K views evenly spaced among them (--views 1: the same view every step)")
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--scene-seed", type=int, default=1)
    ap.add_argument("--slf-res", type=int, default=256)
    ap.add_argument("--layout", type=int, default=0)
    ap.add_argument("--lobes", type=str, default="0,1,2,3,4,5,6", help="0 = diffuse, 1..6 = specular roughness levels")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cfg-5 (path_tracing_single fwd+bwd) measurement appended under `extras`")
    ap.add_argument("--pixel-block", type=int, default=8, help="order valid pixels in BxB image blocks (0 = row-major)")
    ap.add_argument("--per-lobe", action="store_true", help="one launch per lobe (spread over --streams) instead of the single-launch view kernel")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the 7 independent lobe launches of a view are spread over (--per-lobe)")
    ap.add_argument("--emulate-world", type=int, default=0, help="debug: bake only the stripes rank 0 of an N-GPU run would own (no collective), to "
                    "measure the per-rank time of a strong-scaling run on one GPU; the printed value is then NOT the headline metric")
    ap.add_argument("--shard", choices=["stripes", "views"], default=os.environ.get("IRIS_BENCH_SHARD", "stripes"),
                    help="N > 1: 'stripes' = every view sharded over the ranks in interleaved row stripes + one gather per view (BASELINE configs[3]; strong scaling); "
                         "'views' = every rank bakes whole views of the sequence, no collective at all (SURVEY 8(e)'s fallback; weak scaling: N views per step)")
    ap.add_argument("--gather", choices=["gather", "all_gather"], default=os.environ.get("IRIS_BENCH_GATHER", "gather"),
                    help="the one collective of a sharded view: gather to rank 0 (north_star; the rank that writes the files) or all_gather")
    ap.add_argument("--collective-timeout", type=float, default=300.0, help="seconds before a stuck collective aborts the run (a dead rank must not hang the others)")
    ap.add_argument("--long-walls", action="store_true", help="experiments: add the room's six walls once more as 12 large triangles (a decimated scan; see build_workload)")
    ap.add_argument("--emulate-rank", type=int, default=0, help="debug: the rank whose stripes --emulate-world bakes")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE", help="iris_debug_set tuning option (experiments), e.g. bvh_max_leaf=2")
    ap.add_argument("--variant", type=int, default=0, help="bake kernel: 0 auto (tile-sorted), 1 pixel-per-wave, 2 tile-sorted")
    ap.add_argument("--parity-pixels", type=int, default=8192, help="pixels of the last timed view checked bit for bit against the device-arithmetic CPU oracle (0 = skip)")
    ap.add_argument("--cpu-repeats", type=int, default=3, help="repeats of the cpu_baseline sample (the median is reported; each takes --cpu-seconds / repeats)")
    ap.add_argument("--scene-scaling", type=str, default="", help="comma-separated triangle counts: also measure the 7-lobe bake on rooms of these sizes (extras.scene_scaling), e.g. 200000,1000000,3000000,6000000")
    ap.add_argument("--scene-scaling-views", type=int, default=8)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_workers(args.gpus)                     # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # stdout carries exactly ONE line, the JSON: everything else written to file descriptor 1 from here on -- RCCL prints a five-line version banner to the
    # C stdout when its first communicator is created -- goes to stderr; the line itself is written to the saved descriptor at the very end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    assert torch.cuda.is_available(), "bench.py needs the MI355X; there is no CPU fallback"
    # one process per GPU.  (IRIS_BENCH_BACKEND=gloo lets a box with fewer GPUs than ranks exercise the N>1 control flow by
    # sharing devices; it is a functional check only, never a measurement.)
    backend = os.environ.get("IRIS_BENCH_BACKEND", "nccl")
    n_dev = max(torch.cuda.device_count(), 1)
    if backend == "nccl" and world > n_dev:
        if rank == 0:
            print(f"bench.py: --gpus {world} but only {n_dev} HIP device(s) are visible (RCCL needs one GPU per rank)", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank if backend == "nccl" else local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    # IRIS_BENCH_FORCE_PG=1: a world of ONE still creates the process group and sends every view through the collective (RCCL load,
    # dist.gather / all_gather_into_tensor, iris_unstripe_maps on the received buffer) -- the N > 1 code path exercised on a one-GPU box
    force_pg = os.environ.get("IRIS_BENCH_FORCE_PG", "0") == "1"
    have_pg = world > 1 or force_pg
    if have_pg:
        import datetime
        to = datetime.timedelta(seconds=args.collective_timeout)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                    sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=to, world_size=world, rank=rank)   # RCCL over xGMI
        else:
            dist.init_process_group(backend, timeout=to, world_size=world, rank=rank)

    from iris_amd import _lib as L
    from iris_amd import bake_shading as bs
    from iris_amd import sharding as sh
    from iris_amd.utils.dataset import real_ldr
    from tools import synth

    for kv in args.debug_set:
        L.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
    lobes = sorted(int(x) for x in args.lobes.split(","))
    H, W, spp = args.height, args.width, args.spp
    room, slf_np, emi_np, scene, emitter = build_workload(args, dev)
    info = scene.info()
    K, c2w = synth.camera(H, W, 0)
    by_views = world > 1 and args.shard == "views"
    pix_local = sh.local_pixel_ids(H, W, 1 if by_views else world, 0 if by_views else rank, device=dev)
    if args.emulate_world > 1 and world == 1:
        pix_local = sh.local_pixel_ids(H, W, args.emulate_world, args.emulate_rank % args.emulate_world, device=dev)
    rough = bs.roughness_levels().tolist()
    n_maps = (1 if 0 in lobes else 0) + 2 * sum(1 for l in lobes if l > 0)
    one_launch = args.variant == 0 and not args.per_lobe

    gather_stream = torch.cuda.Stream(device=dev) if (have_pg and not by_views) else None
    gatherer = sh.MapGatherer(H, W, world, rank, n_maps, dev, mode=args.gather, force_collective=force_pg) if gather_stream is not None else None     # buffers allocated once per run
    fail_rank, fail_step = int(os.environ.get("IRIS_BENCH_FAIL_RANK", "-1")), int(os.environ.get("IRIS_BENCH_FAIL_STEP", "1"))   # (tests: a rank dying mid-run)
    n_step = [0]
    last = [None]
    last_g = [None]   # the last view's primary-hit tensors (parity_check)
    ev_view = []     # (start, end, rays) HIP events around every bake_view_kernel launch of the timed region, on the launch stream
    ev_gather = []   # the same around the all_gather + permutation (N > 1)

    def step(view=0, timed=False):
        """One view: rays -> primary hits (this rank's stripes) -> all lobes (one launch) -> scatter -> one all_gather."""
        if rank == fail_rank and n_step[0] == fail_step:
            raise RuntimeError(f"IRIS_BENCH_FAIL_RANK: rank {rank} fails in step {fail_step}")
        n_step[0] += 1
        if by_views:
            view = view * world + rank                                          # every rank its own view of the sequence, no exchange
        c2w = synth.camera(H, W, view % args.views, n_views=args.views)[1]      # the train-view sequence: cameras on a circle (cfg 3)
        xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
        xs, ds = xs[pix_local], ds[pix_local]
        g = bs.primary_hits(scene, xs, ds, pixel_ids=pix_local, image_width=W if args.pixel_block else None, block=max(args.pixel_block, 1))
        P = g["position"].shape[0]
        maps = torch.zeros(n_maps, pix_local.numel(), 3, device=dev)
        rays = P * spp * len(lobes)
        if one_launch:
            # default: the whole view behind ONE persistent launch / one tile queue (iris_bake_view)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()          # torch's current stream == the stream the kernel is launched on (L.stream())
            res = bs.bake_lobes(scene, emitter, g["position"], g["normal"], g["wo"], [None if l == 0 else rough[l - 1] for l in lobes], [spp] * len(lobes),
                                seed=0, stream_ids=lobes, pix_id=g["pix_id"])
            if timed:
                e1.record(); ev_view.append((e0, e1, rays))
            pending = list(zip(lobes, res))
        else:
            ls = bs.LobeStreams(dev, args.streams, emitter)
            pending = []
            for l in lobes:
                if l == 0:
                    pending.append((l, ls.run(lambda: bs.bake_diffuse(scene, emitter, g["position"], g["normal"], spp, seed=0, stream_id=0, pix_id=g["pix_id"], variant=args.variant))))
                else:
                    pending.append((l, ls.run(lambda l=l: bs.bake_specular(scene, emitter, g["position"], g["normal"], g["wo"], rough[l - 1], spp, seed=0, stream_id=l,
                                                                            pix_id=g["pix_id"], variant=args.variant))))
            ls.join()
        m = 0
        for l, res in pending:
            if l == 0:
                maps[m, g["sel"]] = res; m += 1
            else:
                maps[m, g["sel"]] = res[0]; maps[m + 1, g["sel"]] = res[1]; m += 2
        if gatherer is not None:
            # the gather of this view runs on its own stream, beside the next view's kernels (bake_shading's CLI hands the maps to its
            # writer threads the same way); the timed region ends with both streams joined
            done = torch.cuda.Event()
            done.record()
            maps.record_stream(gather_stream)
            with torch.cuda.stream(gather_stream):
                gather_stream.wait_event(done)
                if timed:
                    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    g0.record(gather_stream)
                full = gatherer(maps)
                if timed:
                    g1.record(gather_stream); ev_gather.append((g0, g1))
        else:
            full = maps
        last[0] = maps
        last_g[0] = (g, view)
        return rays, full

    def sync():
        if gather_stream is not None:
            torch.cuda.current_stream().wait_stream(gather_stream)
        torch.cuda.synchronize()
        if have_pg:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(view=(i * args.views) // max(args.warmup, 1) + 1)
    sync()
    t0 = time.perf_counter()
    rays_local = 0
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-view times, read after the timed region
    marks[0].record()
    for i in range(args.steps):
        r, full = step(view=(i * args.views) // max(args.steps, 1), timed=True)      # K views evenly spaced over the sequence
        rays_local += r
        marks[i + 1].record()
    sync()
    last_maps = last[0]
    dt_local = time.perf_counter() - t0
    ms_by_view = [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(args.steps)]
    t = torch.tensor([dt_local], device=dev, dtype=torch.float64)
    rays_t = torch.tensor([rays_local], device=dev, dtype=torch.float64)
    ranks_seen = torch.ones(1, device=dev, dtype=torch.float64)
    per_rank = [dt_local]
    gather_ok = None
    if have_pg:
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [float(x.item()) for x in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_t, op=dist.ReduceOp.SUM)
        dist.all_reduce(ranks_seen, op=dist.ReduceOp.SUM)
        if gatherer is not None:
            # the gathered image of the last view holds exactly the ranks' local maps: an order-independent integer checksum (the float bits summed
            # as int64, wrapping) of what every rank sent, reduced over the ranks, against the same checksum of what a receiving rank holds
            sent = last_maps.view(torch.int32).to(torch.int64).sum().reshape(1)
            dist.all_reduce(sent, op=dist.ReduceOp.SUM)
            got = full.view(torch.int32).to(torch.int64).sum().reshape(1) if full is not None else sent.clone()
            bad = (got != sent).to(torch.int64)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            finite = torch.tensor([1 if (full is None or bool(torch.isfinite(full).all())) else 0], device=dev)
            dist.all_reduce(finite, op=dist.ReduceOp.MIN)
            gather_ok = bool(bad.item() == 0) and bool(finite.item() == 1)
    dt = float(t.item()); rays_total = float(rays_t.item())
    value = rays_total / dt / 1e6

    result = {
        "metric": "bake_shading throughput (shading samples/s: secondary rays traced and shaded)", "value": round(value, 2), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3),
        "higher_is_better": True, "scaling": "weak" if by_views else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"bake_shading train-view sequence ({args.steps} views evenly spaced among {args.views} cameras on a circle, one view per step), {W}x{H}, SPP={spp} per lobe, lobes={lobes} (0=diffuse,1-6=specular), synthetic room "
                               f"seed={args.scene_seed} {info['n_triangles']} triangles, SLF H={args.slf_res}, Philox uniforms",
                   "pixels_per_view": H * W, "rays_per_step": int(rays_total / max(args.steps, 1)), "ms_by_view": ms_by_view[:32], "sharding": (f"{world} ranks x whole views of the sequence ({world} views per step), no collective" if by_views else
                                                                                                                        f"{world} x interleaved {sh.STRIPE_ROWS}-row stripes, 1 {args.gather} per view"),
                   "bvh": {"layout": info["layout"], "nodes": info["n_nodes"], "node_bytes": info["node_bytes"], "tri_bytes": info["tri_bytes"], "depth": info["depth"],
                           "sah_cost": round(info["sah_cost"], 3), "build_seconds": round(info["build_seconds"], 2)}},
        "emulated": ({"world": args.emulate_world, "rank": args.emulate_rank % args.emulate_world, "note": "ONE rank's stripes of an N-rank run baked on one GPU, no collective: `value` is NOT the headline metric"}
                     if (args.emulate_world > 1 and world == 1) else None),
        "multi_gpu": {"backend": (("rccl" if backend == "nccl" else backend) if have_pg else None), "process_group": have_pg, "forced_at_world_1": bool(force_pg and world == 1),
                      "ranks_seen_by_all_reduce": int(ranks_seen.item()) if have_pg else None, "per_rank_ms_per_step": [round(x / max(args.steps, 1) * 1e3, 3) for x in per_rank],
                      "gather_ms": round(float(np.mean([a.elapsed_time(b) for a, b in ev_gather])), 3) if ev_gather else None,
                      "gather_overlapped": gatherer is not None, "collective": (args.gather if gatherer is not None else None),
                      "gathered_image_matches_what_the_ranks_sent": gather_ok},
    }

    if rank == 0 and not args.no_roofline and one_launch and ev_view:
        # ---- roofline of the timed kernel (bake_view_kernel: all lobes of a view behind one launch) ----
        ms = [e0.elapsed_time(e1) for e0, e1, _ in ev_view]
        rays_per_launch = float(np.mean([n for _, _, n in ev_view]))
        avg_ms = float(np.mean(ms))
        rate = rays_per_launch / (avg_ms * 1e-3)                       # rays / s inside the kernel
        # (a) per-ray work, live: instrumented launches (include/iris_hip_debug.h) of the same tile code on a pixel sample of view 0, every lobe
        xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
        g = bs.primary_hits(scene, xs[pix_local], ds[pix_local], pixel_ids=pix_local, image_width=W if args.pixel_block else None, block=max(args.pixel_block, 1))
        nP = g["position"].shape[0]
        sel = torch.arange(nP, device=dev)
        sel = sel[(sel // 8192) % 16 == 3] if nP > 16 * 8192 else sel      # every 16th block of 8192 consecutive pixels (whole tiles: real coherence)
        stats = torch.zeros(20, device=dev, dtype=torch.int64)
        for l in lobes:
            if l == 0:
                bs.bake_diffuse(scene, emitter, g["position"][sel], g["normal"][sel], spp, seed=0, stream_id=0, pix_id=g["pix_id"][sel], stats=stats)
            else:
                bs.bake_specular(scene, emitter, g["position"][sel], g["normal"][sel], g["wo"][sel], rough[l - 1], spp, seed=0, stream_id=l, pix_id=g["pix_id"][sel], stats=stats)
        torch.cuda.synchronize()
        st = stats.cpu().numpy().astype(np.float64)
        n_node, n_tri = st[1] / st[0], st[2] / st[0]
        # SURVEY.md section 8(d)'s algorithmic bytes: nodes x 64 B + triangle tests x 48 B + SLF index 4 + radiance row 16 + emitter ordinal 4 + hit-triangle
        # re-read 48, + per-pixel records amortised over spp.  Served by L1 / L2 / Infinity Cache, NOT by HBM: reported, not used as a roof.
        bytes_per_ray = n_node * info["node_bytes"] + n_tri * 48 + (4 + 16 + 4 + 48) + (36 + 4 + 24) / spp
        work = {"nodes_per_ray": round(n_node, 2), "tris_per_ray": round(n_tri, 2), "simd_lane_util_nodes": round(st[1] / max(st[3] * 64, 1), 3),
                "simd_lane_util_tris": round(st[2] / max(st[4] * 64, 1), 3),
                "drain": {"frac_of_node_iterations": round(st[10] / max(st[3], 1), 4), "lane_util": round(st[9] / max(st[10] * 64, 1), 3)},
                "stack_depth_frac_gt_8_12_16": [round(st[5] / st[0], 4), round(st[6] / st[0], 5), round(st[7] / st[0], 6)],
                "top_of_tree_visits_per_ray_lt_21_85_341_1365_nodes": [round(st[k] / st[0], 2) for k in (11, 12, 13, 14)],
                # how much of the node work a wave does TOGETHER (the scalar-top-of-tree question, EXPERIMENTS.md round 4): node steps in which >= 32 lanes sit at one
                # node of one octant table, as a share of all node steps / the visits made in them, as a share of all visits / their SIMD lane utilisation
                "shared_node_steps": {"share_of_node_steps": round(st[15] / max(st[3], 1), 4), "share_of_node_visits": round(st[16] / max(st[1], 1), 4),
                                      "lanes_at_the_shared_node": round(st[16] / max(st[15], 1), 1), "steps_with_every_lane_at_it": round(st[17] / max(st[3], 1), 4),
                                      "share_of_node_steps_executed_through_the_scalar_path": round(st[19] / max(st[3], 1), 4)},
                "note": "per-RAY counts (node visits, triangle tests, stack depths) do not depend on the schedule; the WAVE-level figures (lane utilisation, iterations, drain) come from instrumented "
                        "(COUNT) builds, which since round 5 follow the timed kernel's wave-level schedule (the shared scalar node visits are taken and counted: lanes at other nodes sit such a step "
                        "out, which the lane utilisation of the node steps includes) at 4 instead of 7 waves per SIMD; the timed kernel's lane utilisation over ALL its instructions is "
                        "roofs.valu.simd_lane_utilisation (PMC)"}
        assert 0 <= st[7] <= st[6] <= st[5] <= st[0] and st[3] * 64 >= st[1] and st[19] <= st[3], "instrumented counters violate their invariants"
        # (b) counters of the same kernel from the committed rocprofv3 passes, refused when stale
        pj, src = load_pmc(rays_per_launch, info["node_bytes"])
        roofs, traffic, bound = None, None, None
        if pj:
            c, pr = pj["counters"], pj["rays_per_launch"]
            clock = c["GRBM_GUI_ACTIVE"] / 8 / (pj["duration"]["avg_ns"] * 1e-9)          # shader clock the kernel held in the profiled run (Hz)
            cal = pj["calibration"]
            # VALU issue.  A SIMD issues VALU work in quad-cycles: an instruction holds one (a transcendental two), and gfx950 can pair two `fast`
            # instructions (v_fma_f32, v_mul/add/sub_f32, v_add_u32, v_and/or_b32, v_mov_b32, v_lshrrev_b32: 2 cycles per wave64, 905 G wave-inst/s chip-wide
            # in tools/microbench against 520-590 for everything else) in one.  SQ_ACTIVE_INST_VALU counts the instruction quad-cycles (= SQ_INSTS_VALU +
            # SQ_INSTS_VALU_TRANS_F32 in the profile), SQ_ACTIVE_INST_VALU2 those in which two issued together: their difference is the number of quad-cycles
            # a SIMD's VALU issue was occupied, and the roof is one per SIMD per 4 shader cycles.
            valu_inst_per_ray = c["SQ_INSTS_VALU"] / pr
            quads_per_ray = (c["SQ_ACTIVE_INST_VALU"] - c["SQ_ACTIVE_INST_VALU2"]) / pr
            valu_ach = quads_per_ray * rate / 1e9                                # G issue quad-cycles / s
            valu_peak = N_SIMD * clock / 4 / 1e9
            # vector-memory path (TA / L1): a wave64 load that touches n distinct 64-B lines costs the CU's TA max(9.9, 3.5 + 0.39 n) cycles when it hits L1
            # (tools/microbench gather: 28.1 cycles at 64 lines, 16.1 at 32, 9.9 at <= 16); lines per load = TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD,
            # L1 misses priced at the L2 / Infinity-Cache line rates of the same benchmark.
            loads_per_ray = (c["SQ_INSTS_VMEM_RD"] + c.get("SQ_INSTS_VMEM_WR", 0.0)) / pr
            lines_per_load = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / max(c["SQ_INSTS_VMEM_RD"], 1)
            l1_miss = c["TCP_TCC_READ_REQ_sum"] / max(c["TCP_TOTAL_CACHE_ACCESSES_sum"], 1)
            l2_miss = c["TCC_MISS_sum"] / max(c["TCC_REQ_sum"], 1)
            cyc_line = (1 - l1_miss) * cal["ta_cycles_per_line_l1"] + l1_miss * ((1 - l2_miss) * cal["ta_cycles_per_line_l2"] + l2_miss * cal["ta_cycles_per_line_mall"])
            cyc_load = max(cal["ta_cycles_min_per_load"], cal["ta_cycles_base_per_load"] + cyc_line * lines_per_load)
            ta_ach = loads_per_ray * cyc_load * rate / 1e9                       # G TA cycles / s
            ta_peak = N_CU * clock / 1e9
            # HBM side: bytes that left the L2s (FETCH_SIZE + WRITE_SIZE, KB), against the 8 TB/s HBM peak
            traffic = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            hbm_ach = traffic / pr * rate / 1e9
            prof_rate = pr / (pj["duration"]["avg_ns"] * 1e-9)                    # rays / s of the PROFILED launches (their own duration)
            lane_util = c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"] * 64, 1)
            valu_prof = quads_per_ray * prof_rate / 1e9
            # `frac` of every roof FOLLOWS THIS RUN (advisor round 4): the profiled per-ray counters x THIS run's ray rate inside the kernel (HIP events), against the
            # hardware peak at the 2.4 GHz peak shader clock (a roof is a hardware peak; the clock a box sustains under this kernel is 2.25 ... 2.36 GHz).
            # `frac_profiled` = counters and duration of the SAME profiled launches at THEIR clock: a constant of the profile, kept for recomputation.
            peak_clock = 2.4e9
            valu_peak_hw = N_SIMD * peak_clock / 4 / 1e9
            ta_peak_hw = N_CU * peak_clock / 1e9
            roofs = {
                "valu": {"achieved": round(valu_ach, 1), "peak": round(valu_peak_hw, 1), "unit": "G VALU issue quad-cycles/s", "frac": round(valu_ach / valu_peak_hw, 4),
                         "frac_profiled": round(valu_prof / valu_peak, 4), "frac_profiled_at_2400_MHz": round(valu_prof / valu_peak_hw, 4),
                         "frac_with_this_runs_ray_rate_at_the_profiled_clock": round(valu_ach / valu_peak, 4),
                         "useful_lane_frac": round(valu_ach / valu_peak_hw * lane_util, 4),
                         "frac_note": "frac = profiled issue quad-cycles per ray x THIS run's in-kernel ray rate (HIP events) / the roof at the 2.4 GHz peak clock: it moves with the run.  frac_profiled = counters and duration of the "
                                      "SAME profiled launches at THEIR clock (GRBM_GUI_ACTIVE / duration: the chip lowers its clock under this kernel, differently from box to box; a constant of the profile); "
                                      "frac_profiled_at_2400_MHz = the same against the peak-clock roof; useful_lane_frac = frac x SIMD lane utilisation: the share of the VALU "
                                      "LANE-cycles that did work; pure instruction streams top out at 0.88 (v_fma_f32, dual issue) ... 0.94-0.97 (4- and 8-cycle classes): profiles/r3_counter_calibration.json",
                         "wave_instructions_per_ray": round(valu_inst_per_ray, 1), "issue_quads_per_ray": round(quads_per_ray, 1),
                         "dual_issued_share_of_instructions": round(2 * c["SQ_ACTIVE_INST_VALU2"] / c["SQ_INSTS_VALU"], 3), "profiled_clock_GHz": round(clock / 1e9, 3),
                         "simd_lane_utilisation": round(lane_util, 3), "wave_cycles_waiting": round(c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3)},
                "l1_ta": {"achieved": round(ta_ach, 1), "peak": round(ta_peak_hw, 1), "unit": "G TA cycles/s", "frac": round(ta_ach / ta_peak_hw, 4), "frac_profiled": round(ta_ach * prof_rate / rate / ta_peak, 4),
                          "wave_loads_per_ray": round(loads_per_ray, 2), "lines_per_wave_load": round(lines_per_load, 1), "ta_cycles_per_wave_load": round(cyc_load, 1),
                          "l1_hit": round(1 - l1_miss, 3), "l2_hit": round(1 - l2_miss, 3), "ta_busy_counter": round(c["TA_TA_BUSY_sum"] / N_CU / (c["GRBM_GUI_ACTIVE"] / 8), 3),
                          "td_busy_counter": round(c["TD_TD_BUSY_sum"] / N_CU / (c["GRBM_GUI_ACTIVE"] / 8), 3)},
                "hbm": {"achieved": round(hbm_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_ach / HBM_PEAK_GBS, 4), "frac_profiled": round(hbm_ach * prof_rate / rate / HBM_PEAK_GBS, 4),
                        "bytes_per_ray": round(traffic / pr, 1), "read_bytes_per_ray": round(c["FETCH_SIZE"] * 1024.0 / pr, 1), "write_bytes_per_ray": round(c["WRITE_SIZE"] * 1024.0 / pr, 1),
                        "write_note": "the maps are 0.17 B/ray; the rest of the writes are the per-ray slots (sampled direction, GGX weights, hit) travelling through the workgroup's slab", "note": "raw FETCH_SIZE + WRITE_SIZE: calibrated on known traffic (profiles/r3_counter_calibration.json) FETCH_SIZE reads 1.05 x the bytes of random 64-B record fetches "
                                "(this kernel's pattern) and 0.50 x those of a wide coalesced stream (the guide's case); were every fetch of the stream kind the fraction would be twice this"},
            }
            bound = max(roofs, key=lambda k: roofs[k]["frac"])
        rl = {"kernel": "bake_view_kernel<Q8> (all lobes of a view, one persistent launch)", "launch_ms": round(avg_ms, 3), "launches": len(ms),
              "rays_per_launch": int(rays_per_launch), "mrays_per_s_kernel": round(rate / 1e6, 1), "pmc_source": src,
              "pmc_file": "profiles/" + pj["_file"] if pj else None, "roofs": roofs, "traffic": traffic,
              "algorithmic_bytes_per_ray": round(bytes_per_ray, 1), "algorithmic_GBps_cache_served_exceeds_hbm_peak": round(bytes_per_ray * rate / 1e9, 1),
              "algorithmic_note": "SURVEY 8(d) byte model; these bytes are served by L1 / L2 / Infinity Cache, so the figure exceeds the HBM peak and is not a roof",
              "work_per_ray": work}
        if roofs:
            rl.update({"bound": bound, "achieved": roofs[bound]["achieved"], "peak": roofs[bound]["peak"], "unit": roofs[bound]["unit"], "frac": roofs[bound]["frac"]})
        else:
            rl.update({"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None})
        result["roofline"] = rl

    # ---- the timed output checked: a pixel subsample of the LAST timed view, every lobe, against the CPU oracle in device-arithmetic mode, bit for bit
    parity_failed = False
    if rank == 0 and args.parity_pixels > 0 and one_launch and last_g[0] is not None:
        import oracle
        oracle.build()
        g, view_checked = last_g[0]
        osc = oracle.Scene(room["vertices"], room["faces"])
        oslf = oracle.VoxelSLF(slf_np["inds"], slf_np["radiance"], slf_np["voxel_min"], slf_np["voxel_max"])
        oem = oracle.SLFEmitter(emi_np["is_emitter"], emi_np["emitter_radiance"], emi_np["emitter_area"], oslf)
        P = g["position"].shape[0]
        pick = torch.linspace(0, P - 1, min(args.parity_pixels, P), device=dev).round().long().unique()
        pos, nrm, wo = (g[k][pick].cpu().numpy() for k in ("position", "normal", "wo"))
        pid = g["pix_id"][pick].cpu().numpy().astype(np.int32)
        timed_maps = last_maps[:, g["sel"][pick]].cpu().numpy()            # what the timed step wrote for these pixels (n_maps, n, 3)
        t0c = time.perf_counter()
        mism, m, worst = [], 0, 0
        with oracle.device_arithmetic():
            for l in lobes:
                kw = {} if l == 0 else {"wo": wo, "roughness": np.float32(rough[l - 1])}
                ref = oracle.bake(osc, oem, pos, nrm, spp, seed=0, stream=l, pix_id=pid, **kw)
                for k, r in enumerate(ref):
                    bad = int((timed_maps[m + k].view(np.int32) != r.view(np.int32)).any(1).sum())
                    if bad:
                        mism.append({"lobe": l, "map": k, "pixels": bad})
                    worst = max(worst, bad)
                m += len(ref)
        parity_failed = bool(mism)
        result["parity_check"] = {"bit_exact": not parity_failed, "pixels": int(pick.numel()), "maps": m, "samples_per_pixel_and_lobe": spp, "rays_checked": int(pick.numel()) * spp * len(lobes),
                                  "view": int(view_checked % args.views), "against": "oracle.bake in device-arithmetic mode (oracle/iris_oracle.c, mode 1) on the same primary hits, Philox seed 0",
                                  "what": "the maps written by the LAST TIMED step (not a re-bake), evenly spaced valid pixels", "mismatches": mism[:8], "oracle_seconds": round(time.perf_counter() - t0c, 2)}
        del osc, oem, oslf

    if rank == 0 and world == 1 and not args.no_extras and one_launch:
        # ---- extras with their own events: SURVEY 8(d)'s n_lobes = 1 figure (the diffuse lobe alone: fully incoherent secondary rays) and the reference's own
        # spp mix (bake_shading.py:90,143: diffuse 256, specular 64 / 128 x 5), on views of the same sequence
        def timed_views(lobe_ids, spps, n_views=4):
            ev, rays = [], 0
            for i in range(n_views + 1):
                c2w_i = synth.camera(H, W, (i * args.views) // (n_views + 1), n_views=args.views)[1]
                xs_i, ds_i = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w_i, False, device=dev)
                gi = bs.primary_hits(scene, xs_i[pix_local], ds_i[pix_local], pixel_ids=pix_local, image_width=W if args.pixel_block else None, block=max(args.pixel_block, 1))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                bs.bake_lobes(scene, emitter, gi["position"], gi["normal"], gi["wo"], [None if l == 0 else rough[l - 1] for l in lobe_ids], spps, seed=0, stream_ids=lobe_ids, pix_id=gi["pix_id"])
                e1.record()
                if i:                       # (the first view warms up)
                    ev.append((e0, e1)); rays += gi["position"].shape[0] * sum(spps)
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in ev]
            return {"mrays_per_s": round(rays / (sum(ms) * 1e-3) / 1e6, 1), "kernel_ms_by_view": [round(x, 2) for x in ms], "rays_per_view": int(rays / len(ms)), "lobes": lobe_ids, "spp": spps,
                    "timing": "HIP events around the one bake_view_kernel launch of each view"}
        result.setdefault("extras", {})
        result["extras"]["n_lobes_1_diffuse_only"] = timed_views([0], [spp])
        result["extras"]["reference_spp_mix"] = timed_views([0, 1, 2, 3, 4, 5, 6], [256, 64, 128, 128, 128, 128, 128])

    if rank == 0 and world == 1 and args.scene_scaling:
        # ---- extras.scene_scaling: the 7-lobe bake on rooms of other sizes (the node table exists once per ray octant: 8 x 64 B x nodes)
        import argparse as _ap
        rows = []
        for tris in [int(x) for x in args.scene_scaling.split(",")]:
            a2 = _ap.Namespace(**vars(args)); a2.tris = tris
            t0b = time.perf_counter()
            _, _, _, sc2, em2 = build_workload(a2, dev)
            inf2 = sc2.info()
            ev, rays = [], 0
            for i in range(args.scene_scaling_views + 1):
                c2w_i = synth.camera(H, W, (i * args.views) // (args.scene_scaling_views + 1), n_views=args.views)[1]
                xs_i, ds_i = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w_i, False, device=dev)
                gi = bs.primary_hits(sc2, xs_i[pix_local], ds_i[pix_local], pixel_ids=pix_local, image_width=W if args.pixel_block else None, block=max(args.pixel_block, 1))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                bs.bake_lobes(sc2, em2, gi["position"], gi["normal"], gi["wo"], [None if l == 0 else rough[l - 1] for l in lobes], [spp] * len(lobes), seed=0, stream_ids=lobes, pix_id=gi["pix_id"])
                e1.record()
                if i:
                    ev.append((e0, e1)); rays += gi["position"].shape[0] * spp * len(lobes)
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in ev]
            rows.append({"triangles": inf2["n_triangles"], "nodes": inf2["n_nodes"], "node_table_MB": round(inf2["n_nodes"] * inf2["node_bytes"] * 8 / 1e6, 1), "depth": inf2["depth"],
                         "build_seconds": round(inf2["build_seconds"], 2), "mrays_per_s": round(rays / (sum(ms) * 1e-3) / 1e6, 1), "kernel_ms_by_view": [round(x, 1) for x in ms],
                         "setup_seconds": round(time.perf_counter() - t0b, 1)})
            del sc2, em2
            torch.cuda.empty_cache()
        result.setdefault("extras", {})["scene_scaling"] = rows

    if rank == 0 and world == 1 and not args.no_extras:
        # (before the CPU baseline: its OpenMP team keeps spinning for a while after the last parallel region, and this loop of small launches is
        #  bound by the host thread that issues them -- on a box whose cgroup grants exactly as many CPUs as that team has threads)
        # ---- extras: BASELINE configs[4] (train_brdf_crf / train_emitter inner loop: differentiable one-bounce path tracer, SPP 32) on the same scene
        from tools import bench_pt_single
        # (through the reference's material network -- NGPBRDF as HIP kernels, random parameters of its configuration -- and, for continuity with rounds 1-3,
        #  through the closed-form stand-in those rounds used)
        #  through the closed-form stand-in those rounds used; the stages of a call are timed with HIP events and the two leaders priced: the network against the L2
        #  line roof, the BRDF-sampled rays against this run's own bake-kernel ray rate)
        result.setdefault("extras", {})["cfg5_path_tracing_single"] = bench_pt_single.run(room, slf_np, emi_np, scene, emitter, dev, steps=20, warmup=3, material="ngp", bake_mrays_per_s=result["value"])
        result["extras"]["cfg5_path_tracing_single_network_evaluated_twice_as_the_reference"] = bench_pt_single.run(room, slf_np, emi_np, scene, emitter, dev, steps=20, warmup=3, material="ngp",
                                                                                                                    skip_unused_material=False, stages=False)
        result["extras"]["cfg5_path_tracing_single_stub_material"] = bench_pt_single.run(room, slf_np, emi_np, scene, emitter, dev, steps=20, warmup=3, material="stub", stages=False)
        # A/B rows of the same step (same results as the plain loop, tests/test_pt_single.py): its four independent forward calls issued round-robin on four HIP streams
        # (what a training loop on this chip does: the stages of a 262 144-path call are latency-bound, one call's run beside the next's); the whole step captured as a
        # HIP graph and replayed; and the step as ONE call of spp = SPP = 128 (train_emitter.py:181-189 splits it into SPP // spp calls on the same rays to bound ITS memory:
        # the same estimator, a launch that fills the chip)
        result["extras"]["cfg5_path_tracing_single_calls_on_four_streams"] = bench_pt_single.run(room, slf_np, emi_np, scene, emitter, dev, steps=20, warmup=3, material="ngp", stages=False, streams=4)
        result["extras"]["cfg5_path_tracing_single_hip_graph_replay"] = bench_pt_single.run(room, slf_np, emi_np, scene, emitter, dev, steps=20, warmup=3, material="ngp", stages=False, graph=True)
        result["extras"]["cfg5_path_tracing_single_one_call_of_spp_128"] = bench_pt_single.run(room, slf_np, emi_np, scene, emitter, dev, steps=20, warmup=3, material="ngp", stages=False, spp=128, calls=1)
        # ---- refine_shading's diffuse pass (spp 128, 5 bounces, NEE + MIS) through the same network: the reference's batch and this build's default (16 x)
        from tools import bench_refine
        result["extras"]["refine_diffuse_pass_reference_batch"] = bench_refine.run_f1(scene, emitter, slf_np, dev, batch_pixels=10240, batches=4)
        result["extras"]["refine_diffuse_pass_default_batch"] = bench_refine.run_f1(scene, emitter, slf_np, dev, batch_pixels=163840, batches=2)

    if rank == 0 and args.cpu_seconds > 0:
        # (at N > 1 too: the timed region is over -- `value` is final --, the other ranks wait in the closing barrier, whose timeout is --collective-timeout)
        # ---- CPU baseline: the oracle (port of the same algorithm) on a bounded pixel sample of the same workload ----
        import oracle
        oracle.build()
        osc = oracle.Scene(room["vertices"], room["faces"])
        oslf = oracle.VoxelSLF(slf_np["inds"], slf_np["radiance"], slf_np["voxel_min"], slf_np["voxel_max"])
        oem = oracle.SLFEmitter(emi_np["is_emitter"], emi_np["emitter_radiance"], emi_np["emitter_area"], oslf)
        xs, ds = real_ldr.to_world(real_ldr.get_direction(K, (H, W)), c2w, False, device=dev)
        g = bs.primary_hits(scene, xs, ds)
        pos, nrm, wo = g["position"].cpu().numpy(), g["normal"].cpu().numpy(), g["wo"].cpu().numpy()

        def cpu_run(n_px):
            sel = np.linspace(0, len(pos) - 1, n_px).astype(np.int64)
            t0 = time.perf_counter()
            n = 0
            for l in lobes:
                if l == 0:
                    oracle.bake(osc, oem, pos[sel], nrm[sel], spp, seed=0, stream=0, pix_id=sel.astype(np.int32))
                else:
                    oracle.bake(osc, oem, pos[sel], nrm[sel], spp, wo=wo[sel], roughness=rough[l - 1], seed=0, stream=l, pix_id=sel.astype(np.int32))
                n += n_px * spp
            return n, time.perf_counter() - t0
        # thread count: the host may expose more hardware threads than this job can use (cgroup quota / SMT): every candidate of
        # {all, 1/2, 1/4, 1/8} is calibrated on a short sample and reported; the headline sample runs with the fastest
        ncpu = os.cpu_count() or 1
        try:
            n_aff = len(os.sched_getaffinity(0))
        except Exception:     # noqa
            n_aff = ncpu
        cpu_max = None
        for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            try:
                cpu_max = open(f).read().strip(); break
            except Exception:     # noqa
                pass
        calib = {}
        for th in sorted({max(1, n_aff // d) for d in (1, 2, 4, 8, 16, 32)} | {min(n_aff, 8)}):
            oracle.set_num_threads(th)
            cpu_run(max(th * 8, 64))
            n, dtc = cpu_run(max(th * 48, 512))
            calib[th] = round(n / dtc / 1e6, 3)
        threads = max(calib, key=calib.get)
        oracle.set_num_threads(threads)
        reps = max(1, args.cpu_repeats)
        n_px = int(min(len(pos), max(threads * 8, calib[threads] * 1e6 * args.cpu_seconds / reps / (spp * len(lobes)))))
        runs = [cpu_run(n_px) for _ in range(reps)]                        # repeated, the median reported
        n = runs[0][0]
        times = sorted(r[1] for r in runs)
        dtc = times[len(times) // 2]
        cpu_model = "unknown"
        try:
            cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except Exception:     # noqa
            pass
        cgroup_cpus = None
        try:
            q, per = cpu_max.split()[:2]
            cgroup_cpus = None if q == "max" else round(float(q) / float(per), 2)
        except Exception:     # noqa
            pass
        result["cpu_baseline"] = {"value": round(n / dtc / 1e6, 3), "unit": "Mrays/s", "cores": threads, "threads": threads, "cgroup_cpus": cgroup_cpus, "kind": "port", "cpu_model": cpu_model,
                                  "hw_threads": ncpu, "sched_getaffinity": n_aff, "cgroup_cpu_max": cpu_max, "mrays_per_s_per_thread": round(n / dtc / 1e6 / threads, 4),
                                  "calibration_mrays_per_s_by_threads": calib, "repeats": reps, "seconds_by_repeat": [round(r[1], 2) for r in runs],
                                  "cores_note": "`cores` = OpenMP threads used (the fastest of the calibrated counts); the box's cgroup quota is `cgroup_cpus`",
                                  "sample": f"{n_px} evenly spaced valid pixels of the same view x SPP={spp} x lobes {lobes} = {n} rays, literal (libm) mode, {reps} repeats, median {dtc:.1f} s, OpenMP x{threads}"}
        result["gpu_over_cpu"] = round(value / (n / dtc / 1e6), 1)

    if have_pg:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)
    if parity_failed:
        print("bench.py: parity_check FAILED -- the timed maps differ from the device-arithmetic oracle: " + json.dumps(result["parity_check"]["mismatches"]), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
