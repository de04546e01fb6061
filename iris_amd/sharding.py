"""Multi-GPU sharding of one view (the reference is single-GPU: bake_shading.py:41 pins torch.device(0)).

Every (pixel, sample) is independent, so the path shards with no exchange inside it.  Rows are dealt to ranks in
interleaved stripes (invalid pixels and deep-traversal regions spread evenly); each rank holds a full replica of the
BVH / SLF / emitter tables and bakes only its pixels; ONE collective of the stacked maps (RCCL over xGMI on the GPU box, gloo in the
CPU tests) -- a gather to the rank that writes the files, or an all_gather -- followed by one permutation pass rebuilds the image
(MapGatherer).  Sample streams are
keyed by the image-space pixel id, so the gathered image is bit-identical for every world size."""
import os

import torch
import torch.distributed as dist

# 1080 rows: 135 stripes -> 17 | 16 per rank at N = 8 (max/mean 1.007); 16-row stripes give 144 vs 128 rows (1.067).  IRIS_STRIPE_ROWS: experiments only
# (tools/emulate_ranks.sh; every rank of a job must see the same value)
STRIPE_ROWS = int(os.environ.get("IRIS_STRIPE_ROWS", "8"))
if STRIPE_ROWS < 1:
    raise ValueError(f"IRIS_STRIPE_ROWS={STRIPE_ROWS}: a stripe has at least one row")


def stripe_rows(H, world, rank, stripe=STRIPE_ROWS):
    """Row indices owned by `rank`: stripe s (rows [s*stripe,(s+1)*stripe)) belongs to rank s % world."""
    rows = torch.arange(H)
    return rows[((rows // stripe) % world) == rank]


def local_pixel_ids(H, W, world, rank, stripe=STRIPE_ROWS, device=None):
    """Row-major image-space pixel ids of the rank's stripes (int64, ascending)."""
    rows = stripe_rows(H, world, rank, stripe)
    ids = (rows[:, None] * W + torch.arange(W)[None, :]).reshape(-1)
    return ids.to(device) if device is not None else ids


def max_local_pixels(H, W, world, stripe=STRIPE_ROWS):
    return max(int(stripe_rows(H, world, r, stripe).numel()) for r in range(world)) * W


class MapGatherer:
    """The one collective of a sharded view, with everything it needs allocated ONCE per run (a 1080p view is 13 maps = 323 MB: the send
    buffer, the receive buffer and the image-order output are reused for every view).

      mode "gather"      dist.gather to rank 0 -- north_star's "single RCCL gather": only rank 0 (the rank that writes the files) receives, holds
                         the receive / output buffers and runs the permutation; the other ranks' call returns None
      mode "all_gather"  every rank ends with the full maps (what bench.py's cross-rank check and in-process consumers use)

    The permutation back to image order is one pass: the HIP kernel iris_unstripe_maps for device tensors (pure index arithmetic, no index
    tensors); host tensors (the gloo tests, whose per-rank compute is the oracle) use one cached index per source rank.
    __call__(local_maps (M, n_local, 3) in local_pixel_ids order) -> (M, H*W, 3) or None.  The returned tensor is the gatherer's own buffer:
    it is overwritten by the next call."""

    def __init__(self, H, W, world, rank, n_maps, device, dtype=torch.float32, mode="all_gather", stripe=STRIPE_ROWS, group=None, force_collective=False):
        if mode not in ("gather", "all_gather"):
            raise ValueError("mode must be 'gather' or 'all_gather'")
        self.H, self.W, self.world, self.rank, self.M, self.stripe, self.group, self.mode = H, W, world, rank, n_maps, stripe, group, mode
        self.device = torch.device(device)
        if self.device.type == "cuda" and dtype != torch.float32:
            raise ValueError("MapGatherer: device maps must be float32 (iris_unstripe_maps reads and writes f32)")
        # force_collective: a world of ONE still goes through the send buffer, the collective and the permutation (bench.py IRIS_BENCH_FORCE_PG=1,
        # tests/test_rccl_world1.py: RCCL, dist.gather with a list of views, all_gather_into_tensor and iris_unstripe_maps on an RCCL-written buffer
        # exercised on a one-GPU box)
        self.collective = world > 1 or bool(force_collective)
        if stripe < 1:
            raise ValueError("MapGatherer: stripe height must be >= 1")
        if world > 1:
            # every rank must cut the image into the same stripes (IRIS_STRIPE_ROWS is read per process): a mismatch would send rows the receiver puts elsewhere
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                probe = torch.tensor([stripe, -stripe], dtype=torch.int64, device=self.device if dist.get_backend(group) == "nccl" else "cpu")
                dist.all_reduce(probe, op=dist.ReduceOp.MAX, group=group)
                if int(probe[0]) != stripe or int(-probe[1]) != stripe:
                    raise ValueError(f"MapGatherer: ranks disagree on the stripe height (this rank: {stripe}, max {int(probe[0])}, min {int(-probe[1])})")
        self.n_local = int(stripe_rows(H, world, rank, stripe).numel()) * W
        self.n_max = max_local_pixels(H, W, world, stripe)
        self.receives = self.collective and (mode == "all_gather" or rank == 0)
        self.send = torch.zeros(n_maps, self.n_max, 3, device=self.device, dtype=dtype) if self.collective else None
        self.recv = torch.empty(world, n_maps, self.n_max, 3, device=self.device, dtype=dtype) if self.receives else None
        self.full = torch.empty(n_maps, H * W, 3, device=self.device, dtype=dtype) if self.receives else None
        self._ids = None          # host path: per-rank image pixel ids, built on first use

    def __call__(self, local_maps):
        if not self.collective:
            return local_maps
        M, n_local, C = local_maps.shape
        if (M, n_local, C) != (self.M, self.n_local, 3):
            raise ValueError(f"MapGatherer: expected local maps of shape {(self.M, self.n_local, 3)}, got {tuple(local_maps.shape)}")
        self.send[:, :n_local].copy_(local_maps)                 # (the padding rows stay zero)
        backend = dist.get_backend(self.group)
        if self.mode == "gather":
            dst = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.gather(self.send, [self.recv[r] for r in range(self.world)] if self.rank == 0 else None, dst=dst, group=self.group)
        elif backend == "gloo":
            dist.all_gather([self.recv[r] for r in range(self.world)], self.send, group=self.group)
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        if not self.receives:
            return None
        if self.recv.is_cuda:
            from . import _lib as L
            with torch.cuda.device(self.recv.device):
                L.check(L.lib().iris_unstripe_maps(L.ptr(self.recv), self.world, self.M, self.n_max, self.H, self.W, self.stripe, L.ptr(self.full), L.stream()))
        else:
            if self._ids is None:
                self._ids = [local_pixel_ids(self.H, self.W, self.world, r, self.stripe) for r in range(self.world)]
            for r, ids in enumerate(self._ids):
                self.full.index_copy_(1, ids, self.recv[r, :, : ids.numel()])
        return self.full


def gather_maps(local_maps, H, W, world, rank, stripe=STRIPE_ROWS, group=None, mode="all_gather", force_collective=False):
    """One-shot form of MapGatherer (allocates its buffers for this call; loops over views should keep a MapGatherer).
    local_maps: (M, n_local, 3) rows in local_pixel_ids order -> (M, H*W, 3) full maps (mode "gather": on rank 0, None elsewhere)."""
    if world == 1 and not force_collective:
        return local_maps
    return MapGatherer(H, W, world, rank, local_maps.shape[0], local_maps.device, local_maps.dtype, mode, stripe, group, force_collective)(local_maps)
