"""The bake_shading stage (reference: bake_shading.py) on MI355X.

``bake_diffuse`` / ``bake_specular`` are the reference's two hot loops (bake_shading.py:108-123, :168-188) as ONE
kernel launch each: uniforms, BRDF sampling, secondary-ray traversal, emitter / SLF lookup, weights and the
mean over spp never leave the chip.  ``bake_view`` is one iteration of the reference's per-view loops
(:93-131 and :149-204) and returns the 13 pre-denoise maps; ``main`` keeps the reference's CLI.
"""
import math
import os
import time

import torch

from . import _lib as L
from .utils.path_tracing import ray_intersect

SPP_DIFFUSE = 256                                  # bake_shading.py:90
SPPS_SPECULAR = [64, 128, 128, 128, 128, 128]      # bake_shading.py:143
N_ROUGHNESS = 6                                    # bake_shading.py:147


def roughness_levels():
    """torch.linspace(0.02,1.0,6) (bake_shading.py:147)"""
    return torch.linspace(0.02, 1.0, N_ROUGHNESS)


def _common(scene, emitter, position, normal, spp, u2, pix_id, want_tri):
    position = L.require_gpu(position, torch.float32, "position").reshape(-1, 3)
    normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
    P = position.shape[0]
    dev = position.device
    if u2 is not None:
        u2 = L.require_gpu(u2, torch.float32, "u2").reshape(-1, 2)
        if u2.shape[0] != P * spp:
            raise L.IrisError(f"u2 must have P*spp = {P * spp} rows, got {u2.shape[0]}")
    if pix_id is not None:
        pix_id = L.require_gpu(pix_id, torch.int32, "pix_id").reshape(-1)
        if pix_id.shape[0] != P:
            raise L.IrisError("pix_id must have one entry per pixel")
    tri = torch.empty(P * spp, device=dev, dtype=torch.int64) if want_tri else None
    return position, normal, P, dev, u2, pix_id, tri


def _workspace(P, spp, specular, variant, dev):
    """Scratch of the tile-sorted kernel (per-ray results between its trace and reduce phases); a torch allocation so
    that it is stream-ordered like every other tensor."""
    if variant == L.BAKE_PIXEL_PER_WAVE:
        return None, 0
    n = int(L.lib().iris_bake_workspace_bytes(P, int(spp), int(specular)))
    if n == 0:
        return None, 0
    return torch.empty(n, device=dev, dtype=torch.uint8), n


def bake_diffuse(scene, emitter, position, normal, spp=SPP_DIFFUSE, u2=None, seed=0, stream_id=0, pix_id=None, want_tri=False, want_src=False,
                 stats=None, variant=L.BAKE_AUTO):
    """Ld_ of bake_shading.py:108-123 for all P valid pixels: mean over spp of Le along cosine-sampled rays.
    u2: optional (P*spp,2) uniforms in the reference's order (parity mode); otherwise in-kernel Philox.
    want_tri: also return the per-sample hit triangle (P*spp,) int64.
    Diagnostics (routed through include/iris_hip_debug.h): want_src -> per-sample radiance-table row; stats -> zeroed int64[20]
    device tensor for an instrumented launch; variant -> a specific kernel.  Returns Ld [, tri] [, src]."""
    position, normal, P, dev, u2, pix_id, tri = _common(scene, emitter, position, normal, spp, u2, pix_id, want_tri)
    Ld = torch.empty(P, 3, device=dev, dtype=torch.float32)
    src = torch.empty(P * spp, device=dev, dtype=torch.int64) if want_src else None
    ws, ws_bytes = _workspace(P, spp, False, variant, dev)
    with torch.cuda.device(dev):
        if want_src or stats is not None or variant != L.BAKE_AUTO:
            L.check(L.lib().iris_debug_bake_diffuse(scene.handle, emitter.handle(dev), emitter.slf.handle(dev), L.ptr(position), L.ptr(normal),
                                                    P, int(spp), L.ptr(u2), int(seed), int(stream_id), L.ptr(pix_id), L.ptr(Ld), L.ptr(tri), L.ptr(src),
                                                    L.ptr(stats), int(variant), L.ptr(ws), ws_bytes, L.stream()))
        else:
            L.check(L.lib().iris_bake_diffuse(scene.handle, emitter.handle(dev), emitter.slf.handle(dev), L.ptr(position), L.ptr(normal),
                                              P, int(spp), L.ptr(u2), int(seed), int(stream_id), L.ptr(pix_id), L.ptr(Ld), L.ptr(tri),
                                              L.ptr(ws), ws_bytes, L.stream()))
    res = (Ld,) + ((tri,) if want_tri else ()) + ((src,) if want_src else ())
    return res if len(res) > 1 else Ld


def bake_specular(scene, emitter, position, normal, wo, roughness, spp, u2=None, seed=0, stream_id=1, pix_id=None, want_tri=False, want_src=False,
                  stats=None, variant=L.BAKE_AUTO):
    """Ls0_, Ls1_ of bake_shading.py:168-188 for one roughness level.  Returns Ls0, Ls1 [, tri] [, src] (see bake_diffuse)."""
    position, normal, P, dev, u2, pix_id, tri = _common(scene, emitter, position, normal, spp, u2, pix_id, want_tri)
    wo = L.require_gpu(wo, torch.float32, "wo").reshape(-1, 3)
    if isinstance(roughness, torch.Tensor):
        roughness = float(roughness.detach().float().cpu().item())
    Ls0 = torch.empty(P, 3, device=dev, dtype=torch.float32)
    Ls1 = torch.empty(P, 3, device=dev, dtype=torch.float32)
    src = torch.empty(P * spp, device=dev, dtype=torch.int64) if want_src else None
    ws, ws_bytes = _workspace(P, spp, True, variant, dev)
    with torch.cuda.device(dev):
        if want_src or stats is not None or variant != L.BAKE_AUTO:
            L.check(L.lib().iris_debug_bake_specular(scene.handle, emitter.handle(dev), emitter.slf.handle(dev), L.ptr(position), L.ptr(normal), L.ptr(wo),
                                                     roughness, P, int(spp), L.ptr(u2), int(seed), int(stream_id), L.ptr(pix_id), L.ptr(Ls0), L.ptr(Ls1),
                                                     L.ptr(tri), L.ptr(src), L.ptr(stats), int(variant), L.ptr(ws), ws_bytes, L.stream()))
        else:
            L.check(L.lib().iris_bake_specular(scene.handle, emitter.handle(dev), emitter.slf.handle(dev), L.ptr(position), L.ptr(normal), L.ptr(wo),
                                               roughness, P, int(spp), L.ptr(u2), int(seed), int(stream_id), L.ptr(pix_id), L.ptr(Ls0), L.ptr(Ls1),
                                               L.ptr(tri), L.ptr(ws), ws_bytes, L.stream()))
    return (Ls0, Ls1) + ((tri,) if want_tri else ()) + ((src,) if want_src else ())


def bake_lobes(scene, emitter, position, normal, wo, roughness, spps, seed=0, stream_ids=None, pix_id=None):
    """All lobes of one view in ONE launch (`iris_bake_view`): roughness[l] is None for the diffuse lobe, a float otherwise.
    Returns a list with Ld (P,3) for diffuse entries and (Ls0, Ls1) for specular ones -- bit-identical to bake_diffuse /
    bake_specular with the same stream ids (default: 0 for diffuse, 1 + position in linspace(0.02,1,6) otherwise)."""
    import ctypes as C
    position = L.require_gpu(position, torch.float32, "position").reshape(-1, 3)
    normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
    wo = L.require_gpu(wo, torch.float32, "wo").reshape(-1, 3)
    P, dev, n = position.shape[0], position.device, len(roughness)
    if pix_id is not None:
        pix_id = L.require_gpu(pix_id, torch.int32, "pix_id").reshape(-1)
    levels = roughness_levels().tolist()
    if stream_ids is None:
        stream_ids = [0 if r is None else 1 + min(range(N_ROUGHNESS), key=lambda k: abs(levels[k] - float(r))) for r in roughness]
    outs0 = [torch.empty(P, 3, device=dev, dtype=torch.float32) for _ in range(n)]
    outs1 = [None if roughness[l] is None else torch.empty(P, 3, device=dev, dtype=torch.float32) for l in range(n)]
    rough = (C.c_float * n)(*[-1.0 if r is None else float(r) for r in roughness])
    spp_a = (C.c_int32 * n)(*[int(s) for s in spps])
    sid = (C.c_uint32 * n)(*[int(s) for s in stream_ids])
    p0 = (C.c_void_p * n)(*[t.data_ptr() for t in outs0])
    p1 = (C.c_void_p * n)(*[None if t is None else t.data_ptr() for t in outs1])
    ws, ws_bytes = _workspace(P, 1, True, L.BAKE_AUTO, dev)
    with torch.cuda.device(dev):
        L.check(L.lib().iris_bake_view(scene.handle, emitter.handle(dev), emitter.slf.handle(dev), L.ptr(position), L.ptr(normal), L.ptr(wo), L.ptr(pix_id),
                                       P, n, rough, spp_a, sid, int(seed), p0, p1, L.ptr(ws), ws_bytes, L.stream()))
    return [outs0[l] if roughness[l] is None else (outs0[l], outs1[l]) for l in range(n)]


def view_seed(seed, im_id):
    """Philox key of view `im_id` for a run seeded with `seed` (64-bit golden-ratio mix; im_id 0 keeps the plain seed)."""
    return (int(seed) + int(im_id) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


class LobeStreams:
    """Runs the independent lobe launches of one view round-robin on a few HIP streams, so that the tail of one persistent
    kernel (its last tiles) overlaps with the head of the next instead of idling the chip; `join()` makes the caller's stream
    wait for all of them.  With n=1 everything stays on the caller's stream."""

    def __init__(self, device, n=3, emitter=None):
        self.main = torch.cuda.current_stream(device)
        if emitter is not None:
            # a pending re-upload of the radiance tables (in-place edit, load_state_dict) happens HERE, on the caller's stream, which every side
            # stream waits for -- not inside the first lobe's side stream, where the other lobes' kernels would not be ordered after it
            emitter.handle(device); emitter.slf.handle(device)
        self.side = [torch.cuda.Stream(device=device) for _ in range(n)] if n > 1 else []
        self.k = 0

    def run(self, fn):
        if not self.side:
            return fn()
        st = self.side[self.k % len(self.side)]
        self.k += 1
        st.wait_stream(self.main)                 # inputs were produced on the caller's stream
        with torch.cuda.stream(st):
            return fn()                           # outputs / workspace are allocated in this stream's pool

    def join(self):
        for st in self.side:
            self.main.wait_stream(st)


_BLOCK_ORDER = {}     # (rows, id(pixel_ids), image_width, block, device) -> (pixel_ids kept alive, its version, permutation): a view sequence reuses one entry


def _block_order(n, pixel_ids, image_width, block, device):
    """Permutation of the rows 0..n-1 (image pixel pixel_ids[row], or row itself) that lists them block x block image tile by tile, row-major inside a
    tile.  Depends only on the camera raster (and the rank's stripes), so it is computed once and kept."""
    key = (n, None if pixel_ids is None else id(pixel_ids), image_width, block, str(device))
    hit = _BLOCK_ORDER.get(key)
    if hit is not None and (pixel_ids is None or (hit[0] is pixel_ids and hit[1] == pixel_ids._version)):
        return hit[2]
    pix = torch.arange(n, device=device) if pixel_ids is None else pixel_ids.to(device=device, dtype=torch.int64)
    y, x = pix // image_width, pix % image_width
    nbx = (image_width + block - 1) // block
    perm = torch.argsort(((y // block) * nbx + x // block) * (block * block) + (y % block) * block + x % block)
    if len(_BLOCK_ORDER) >= 8:
        _BLOCK_ORDER.pop(next(iter(_BLOCK_ORDER)))
    _BLOCK_ORDER[key] = (pixel_ids, None if pixel_ids is None else pixel_ids._version, perm)
    return perm


def primary_hits(scene, xs, ds, pixel_ids=None, image_width=None, block=8):
    """bake_shading.py:98-101 / :154-157: primary closest hit + compaction to the valid pixels.
    Returns dict(position, normal, wo, pix_id (int32 image-space pixel index), sel, n_pixels).
    image_width: when given, the valid pixels are ordered in `block` x `block` image blocks instead of row-major, so that the
    consecutive pixels the bake kernels group into a tile are neighbours in 2-D (nearer origins, more coherent rays).  The
    order of the pixel list changes neither the results (sample streams are keyed by pix_id) nor the caller's layout
    (`sel` scatters the rows back)."""
    positions, normals, _, _, valid = ray_intersect(scene, xs, ds)
    if image_width is not None:
        # rows in block order, then the valid ones among them: the same list as sorting the valid rows by block key, without a sort per view
        perm = _block_order(xs.shape[0], pixel_ids, image_width, block, xs.device)
        sel = perm[valid[perm]]
    else:
        sel = torch.nonzero(valid, as_tuple=False).reshape(-1)
    pix = sel if pixel_ids is None else pixel_ids[sel]
    return {"position": positions[sel], "normal": normals[sel], "wo": -ds.reshape(-1, 3)[sel],
            "pix_id": pix.to(torch.int32), "sel": sel, "n_pixels": xs.shape[0]}


def bake_view(scene, emitter, xs, ds, spp_diffuse=SPP_DIFFUSE, spps_specular=None, seed=0, pixel_ids=None, lobes=None,
              image_width=None, n_streams=3, denoiser=None):
    """One view: the primary pass once (the reference repeats it, :98 and :154), then the diffuse lobe and the six
    specular roughness levels.  Returns {'diffuse': (N,3), 'specular0': [6x (N,3)], 'specular1': [...], 'n_valid', 'rays'}
    with N = len(xs) rows in the caller's pixel order (zeros at invalid pixels, bake_shading.py:126-127).
    denoiser: a utils.denoise.Denoiser for the full image (xs must then be all H*W pixels in image order): filters the diffuse map
    and the specular maps of roughness levels > 0 as the reference does (:129, :198-200), guided by this view's primary hits."""
    spps = list(SPPS_SPECULAR if spps_specular is None else spps_specular)
    g = primary_hits(scene, xs, ds, pixel_ids, image_width=image_width)
    N, dev = xs.shape[0], xs.device
    out = {"n_valid": int(g["sel"].shape[0]), "rays": 0, "specular0": [], "specular1": []}
    P = out["n_valid"]

    def scatter(v):
        img = torch.zeros(N, 3, device=dev, dtype=torch.float32)
        img[g["sel"]] = v
        return img
    levels = roughness_levels().tolist()
    want = [l for l in range(N_ROUGHNESS + 1) if lobes is None or l in lobes]
    spp_of = lambda l: spp_diffuse if l == 0 else spps[l - 1]
    if P > 0 and want and all(spp_of(l) <= int(L.lib().iris_bake_tile_max_spp()) for l in want):
        # one persistent launch for the whole view
        res = bake_lobes(scene, emitter, g["position"], g["normal"], g["wo"], [None if l == 0 else levels[l - 1] for l in want], [spp_of(l) for l in want],
                         seed=seed, stream_ids=want, pix_id=g["pix_id"])
        pending = list(zip(want, res))
    else:
        ls = LobeStreams(dev, n_streams, emitter)
        pending = []
        for l in want:
            if l == 0:
                pending.append((0, ls.run(lambda: bake_diffuse(scene, emitter, g["position"], g["normal"], spp_diffuse, seed=seed, stream_id=0, pix_id=g["pix_id"]))))
            else:
                pending.append((l, ls.run(lambda l=l: bake_specular(scene, emitter, g["position"], g["normal"], g["wo"], levels[l - 1], spps[l - 1], seed=seed,
                                                                    stream_id=l, pix_id=g["pix_id"]))))
        ls.join()
    for lobe, res in pending:
        out["rays"] += P * spp_of(lobe)
        if lobe == 0:
            out["diffuse"] = scatter(res)
        else:
            out["specular0"].append(scatter(res[0])); out["specular1"].append(scatter(res[1]))
    if denoiser is not None:
        if pixel_ids is not None or N != denoiser.H * denoiser.W:
            raise L.IrisError("bake_view: denoising needs the whole image in image order")
        valid = torch.zeros(N, dtype=torch.bool, device=dev); valid[g["sel"]] = True
        denoiser.set_guides(scatter(g["normal"]), scatter(g["position"]), valid)
        names = [("diffuse", None)] if "diffuse" in out else []
        lv = [l for l in want if l > 1]                    # "no need for denoise of low roughness" (:198): level index 0 stays as baked
        order = [l for l in want if l > 0]
        names += [(k, order.index(l)) for l in lv for k in ("specular0", "specular1")]
        den = denoiser.denoise_maps([out[k] if i is None else out[k][i] for k, i in names])
        for (k, i), d in zip(names, den):
            if i is None:
                out[k] = d.reshape(N, 3)
            else:
                out[k][i] = d.reshape(N, 3)
    return out


# ------------------------------------------------------------------------------------------------------------------
# CLI: same flags and output files as the reference's bake_shading.py (:29-39, :131, :202-203)
# ------------------------------------------------------------------------------------------------------------------
def output_files(output, im_id):
    d = [os.path.join(output, "diffuse", "{:03d}.exr".format(im_id))]
    s = [os.path.join(output, "specular", "{:03d}_{}_{}.exr".format(im_id, k, r)) for r in range(N_ROUGHNESS) for k in (0, 1)]
    return d + s


class MapWriter:
    """The 13 maps of a view -> host -> 13 EXR files, off the critical path (bake_shading.py:131,202-203 write them synchronously).
    submit() lays the maps out as the file stores them on the GPU (B,G,R planes per scanline block, ZIP's byte reordering + delta
    predictor: utils/exr.scanline_blocks_torch), enqueues one device-to-host copy into pinned buffers on a side stream and returns; writer
    threads wait for the copy's event, deflate and write, so the next view's kernels start while this view is still on its way to disk and
    the host threads do nothing under the GIL but file I/O.  Two buffer sets: a third view waits for the first one's files.  Files appear
    under their final name only when complete (the CLI's resume rule never sees a truncated file)."""

    def __init__(self, device, img_hw, compression, n_maps=13, n_buffers=2):
        from concurrent.futures import ThreadPoolExecutor
        from .utils import exr
        self.exr, self.compression, self.img_hw, self.n_buffers = exr, compression, tuple(img_hw), n_buffers
        self.stream = torch.cuda.Stream(device=device)
        self.bufs = None                                    # pinned (full, tail) pairs, sized at the first submit
        self.busy = [[] for _ in range(n_buffers)]
        self.pool = ThreadPoolExecutor(max_workers=max(1, min(n_maps, os.cpu_count() or 4)), thread_name_prefix="maps")
        self.k = 0

    def submit(self, files, maps_dev, on_written=None):
        """maps_dev: (n_maps, H, W, 3) f32 on the device (R,G,B), produced on the current stream; files: n_maps paths.
        on_written: called (on a writer thread) once ALL of this view's files are complete on disk -- refine_shading's per-view resume marker."""
        exr, (H, W) = self.exr, self.img_hw
        full, tail = exr.scanline_blocks_torch(maps_dev, self.compression)
        if self.bufs is None:
            self.bufs = [(torch.empty(full.shape, dtype=torch.uint8).pin_memory(), torch.empty(tail.shape, dtype=torch.uint8).pin_memory())
                         for _ in range(self.n_buffers)]
        i = self.k % self.n_buffers; self.k += 1
        for f in self.busy[i]:
            f.result()                                      # this buffer's previous files are on disk (re-raises a writer's exception)
        ready = torch.cuda.Event(); ready.record()
        done = torch.cuda.Event()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            self.bufs[i][0].copy_(full, non_blocking=True)
            self.bufs[i][1].copy_(tail, non_blocking=True)
            done.record(self.stream)
        full.record_stream(self.stream); tail.record_stream(self.stream)
        hfull, htail = self.bufs[i][0].numpy(), self.bufs[i][1].numpy()
        chunks = exr.chunk_pool() if (os.cpu_count() or 1) > 16 else None     # more cores than files in flight: deflate block-parallel

        def write(path, j):
            done.synchronize()
            tmp = path + ".part"
            exr.write_exr_blocks(tmp, H, W, self.compression, hfull[j], htail[j], pool=chunks)
            os.replace(tmp, path)
        writes = [self.pool.submit(write, f, j) for j, f in enumerate(files)]
        self.busy[i] = list(writes)
        if on_written is not None:
            def finish(writes=tuple(writes)):               # (bound now: the list below also holds this task's own future)
                for f in writes:
                    f.result()                              # (a failed write re-raises here and the marker is not written)
                on_written()
            self.busy[i].append(self.pool.submit(finish))   # queued behind the writes it waits for

    def close(self):
        for fs in self.busy:
            for f in fs:
                f.result()
        self.busy = [[] for _ in range(self.n_buffers)]
        self.pool.shutdown()


def main(argv=None):
    from argparse import ArgumentParser
    from .model.emitter import SLFEmitter
    from .utils import cameras, exr
    from .utils.path_tracing import load_scene
    parser = ArgumentParser(description="bake diffuse / specular shading maps (MI355X)")
    parser.add_argument("--dataset_root", type=str, help="dataset root")
    parser.add_argument("--scene", type=str, required=True, help="dataset folder")
    parser.add_argument("--slf_path", type=str, required=True)
    parser.add_argument("--emitter_path", type=str, required=True)
    parser.add_argument("--output", type=str, required=True, help="output path")
    parser.add_argument("--dataset", type=str, required=True, help="dataset type: synthetic | real | scannetpp | generic")
    parser.add_argument("--ldr_img_dir", type=str, default=None)       # accepted for compatibility; bake never reads images
    parser.add_argument("--res_scale", type=float, default=1.0)
    # additions (defaults reproduce the reference)
    parser.add_argument("--cameras", type=str, default=None, help="generic camera JSON instead of the dataset's own camera files")
    parser.add_argument("--img_hw", type=int, nargs=2, default=None)
    parser.add_argument("--spp_diffuse", type=int, default=SPP_DIFFUSE)
    parser.add_argument("--spps_specular", type=int, nargs=N_ROUGHNESS, default=SPPS_SPECULAR)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--compression", type=str, default="zip", choices=["none", "zips", "zip"])
    parser.add_argument("--overwrite", action="store_true", help="re-bake views whose 13 files already exist")
    parser.add_argument("--denoise", type=str, default="atrous", choices=["atrous", "none"],
                        help="atrous: guided a-trous filter in place of the reference's OptiX denoiser (:129, :198-200); none: raw Monte-Carlo maps")
    args = parser.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise L.IrisError("bake_shading needs a HIP device; there is no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if args.dataset in ("synthetic", "real"):
        mesh_path = os.path.join(args.scene, "scene.obj")
    elif args.dataset == "scannetpp":
        mesh_path = os.path.join(args.dataset_root, "data", args.scene, "scans", "scene.ply")
    else:
        mesh_path = os.path.join(args.scene, "scene.obj") if os.path.exists(os.path.join(args.scene, "scene.obj")) else os.path.join(args.scene, "scene.ply")
    assert os.path.exists(mesh_path), "mesh not found: " + mesh_path
    scene = load_scene(mesh_path, device=device)

    hw = tuple(args.img_hw) if args.img_hw else None
    if args.cameras:
        img_hw, views = cameras.load_generic(args.cameras, args.res_scale)
    elif args.dataset == "synthetic":
        img_hw, views = cameras.load_synthetic(args.scene, args.res_scale, hw)
    elif args.dataset == "real":
        img_hw, views = cameras.load_real(args.scene, args.res_scale, hw)
    elif args.dataset == "scannetpp":                      # Scannetpp(args.dataset_root, args.scene, split='train', pixel=False, res_scale=args.res_scale)
        img_hw, views = cameras.load_scannetpp(args.dataset_root, args.scene, args.res_scale)
    else:
        raise L.IrisError("--dataset {!r}: synthetic | real | scannetpp, or --cameras cameras.json".format(args.dataset))

    emitter = SLFEmitter(args.emitter_path, args.slf_path)
    for p in emitter.parameters():
        p.requires_grad = False
    os.makedirs(os.path.join(args.output, "diffuse"), exist_ok=True)
    os.makedirs(os.path.join(args.output, "specular"), exist_ok=True)

    denoiser = None
    if args.denoise == "atrous":
        from .utils.denoise import Denoiser
        denoiser = Denoiser(img_hw[::-1], device)          # denoiser = mitsuba.OptixDenoiser(img_hw[::-1])   (:81)
    writer = MapWriter(device, img_hw, args.compression)
    start_time = time.time()
    rays = 0
    for im_id in range(rank, len(views), world):           # views shard over ranks with no collective: one file set per view
        files = output_files(args.output, im_id)
        if not args.overwrite and all(os.path.exists(f) for f in files):
            continue
        marker = os.path.join(args.output, "diffuse", "{:03d}.refined".format(im_id))
        if os.path.exists(marker):
            os.remove(marker)                               # the view is baked again: refine_shading --resume must not take the new files for refined ones
        xs, ds = cameras.view_rays(views[im_id], img_hw, device)
        # per-view Philox key: the reference's torch.rand stream advances from view to view, so its Monte-Carlo noise is independent
        # across the training views; keyed on the view id (not on the rank), so that results do not depend on how views are sharded
        out = bake_view(scene, emitter, xs, ds, args.spp_diffuse, args.spps_specular, seed=view_seed(args.seed, im_id), image_width=img_hw[1], denoiser=denoiser)
        rays += out["rays"]
        # 13 maps -> pinned host buffer on a side stream -> compressed and written by threads while the next view bakes (a 1080p ZIP map
        # costs ~1 s of CPU, the bake of the whole view 0.3 s of GPU)
        writer.submit(files, torch.stack([out["diffuse"]] + [out[k][r] for r in range(N_ROUGHNESS) for k in ("specular0", "specular1")]).reshape(13, *img_hw, 3))
    writer.close()
    torch.cuda.synchronize()
    dt = time.time() - start_time
    print("[bake_shading] rank {}: {} rays in {:.2f} s ({:.1f} Mrays/s incl. file I/O)".format(rank, rays, dt, rays / max(dt, 1e-9) / 1e6))


if __name__ == "__main__":
    main()
