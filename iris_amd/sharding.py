"""Multi-GPU sharding of one view (the reference is single-GPU: bake_shading.py:41 pins torch.device(0)).

Every (pixel, sample) is independent, so the path shards with no exchange inside it.  Rows are dealt to ranks in
interleaved stripes (invalid pixels and deep-traversal regions spread evenly); each rank holds a full replica of the
BVH / SLF / emitter tables and bakes only its pixels; ONE all_gather of the stacked maps (RCCL over xGMI on the GPU
box, gloo in the CPU tests) followed by a local permutation rebuilds the image on every rank.  Sample streams are
keyed by the image-space pixel id, so the gathered image is bit-identical for every world size."""
import torch
import torch.distributed as dist

STRIPE_ROWS = 8     # 1080 rows: 135 stripes -> 17 | 16 per rank at N = 8 (max/mean 1.007); 16-row stripes give 144 vs 128 rows (1.067)


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


def gather_maps(local_maps, H, W, world, rank, stripe=STRIPE_ROWS, group=None):
    """local_maps: (M, n_local, 3) rows in local_pixel_ids order -> (M, H*W, 3) full maps on every rank.
    One collective: all_gather_into_tensor of equally padded buffers, then an index_copy per source rank."""
    M, n_local, C = local_maps.shape
    if world == 1:
        return local_maps
    n_max = max_local_pixels(H, W, world, stripe)
    buf = torch.zeros(M, n_max, C, device=local_maps.device, dtype=local_maps.dtype)
    buf[:, :n_local] = local_maps
    out = torch.empty(world, M, n_max, C, device=local_maps.device, dtype=local_maps.dtype)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        out = torch.stack(parts)
    else:
        dist.all_gather_into_tensor(out, buf, group=group)
    full = torch.zeros(M, H * W, C, device=local_maps.device, dtype=local_maps.dtype)
    for r in range(world):
        ids = local_pixel_ids(H, W, world, r, stripe, device=local_maps.device)
        full.index_copy_(1, ids, out[r, :, : ids.numel()])
    return full
