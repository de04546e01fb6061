"""What the parity tests and the fixture generators share: the workloads of the room-scale and configs[1]-size fixtures (scene / SLF / emitters / camera
from tools/synth.py -- identical in the build container and on the GPU box), the lossless delta codec the configs[1] fixture is stored with, the per-pixel
sample hash and the error measure.  Nothing here touches /root/reference: the generators that do (tools/make_*golden*.py) stay in the build container
(.gpurunignore); this module travels with the tests.
"""
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# tests/golden/bake_room.npz: room scale
ROOM = dict(H=120, W=160, SPP=64, VIEW=3, TRIS=200_000, SLF_H=256, SEED=0, SCENE_SEED=0)
# tests/golden/bake_cfg2_reference.npz: BASELINE configs[1] size on the bench scene
CFG2 = dict(H=480, W=640, SPP=64, VIEW=0, TRIS=1_000_000, SLF_H=256, SEED=0, SCENE_SEED=1)


def workload(cfg):
    """Scene / SLF / emitter tables / camera of a fixture (numpy)."""
    from tools import synth
    room = synth.room(cfg["SCENE_SEED"], cfg["TRIS"])
    slf = synth.slf_for(room["vertices"], room["faces"], cfg["SLF_H"])
    emi = synth.emitters_for(room["vertices"], room["faces"], room["is_emitter"])
    K, c2w = synth.camera(cfg["H"], cfg["W"], cfg["VIEW"])
    return room, slf, emi, K, c2w


def sample_hash(tri, src, P, spp):
    """64-bit hash per pixel of its samples' (triangle id, radiance-table row) sequence (uint64 wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        t = (np.asarray(tri, np.int64).astype(np.uint64) + np.uint64(2)) * np.uint64(0x9E3779B97F4A7C15)
        s = (np.asarray(src, np.int64).astype(np.uint64) + np.uint64(1 << 40)) * np.uint64(0xC2B2AE3D27D4EB4F)
        k = (np.arange(spp, dtype=np.uint64) * np.uint64(2) + np.uint64(1))[None, :]
        v = (t ^ (s >> np.uint64(7)) ^ (s << np.uint64(13))).reshape(P, spp) * k
        return v.sum(1, dtype=np.uint64)


def rel(a, b, mask=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if mask is not None:
        a, b = a[mask], b[mask]
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def ulp_key(a):
    """float32 -> int64 whose differences count representable floats (monotone in the value; -0 and +0 adjacent)"""
    i = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    return np.where(i < 0, -(i & 0x7FFFFFFF) - 1, i)      # -0.0 -> -1, +0.0 -> 0: adjacent, monotone


def delta_encode(ref, pred):
    """ref, pred float32 arrays of one shape -> (int8 ulp deltas with 0 at the exceptions, exception flat indices int32, exception values f32)"""
    d = ulp_key(ref) - ulp_key(pred)
    exc = (np.abs(d) > 127) | ~np.isfinite(ref) | ~np.isfinite(pred)
    d8 = np.where(exc, 0, d).astype(np.int8)
    idx = np.nonzero(exc.reshape(-1))[0].astype(np.int32)
    return d8, idx, np.ascontiguousarray(ref, np.float32).reshape(-1)[idx]


def delta_decode(pred, d8, idx, val):
    """inverse of delta_encode: the reference array, bit for bit"""
    k = ulp_key(pred) + d8.astype(np.int64)
    i = np.where(k < 0, (-(k + 1)) | 0x80000000, k).astype(np.uint32)
    out = i.view(np.float32).reshape(pred.shape).copy()
    out.reshape(-1)[idx] = val
    return out


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# tests/golden/pt_single_cfg5_reference.npz: BASELINE configs[4] size (train_emitter's inner loop: 8 192 pixels x spp 32 per call) on the bench scene
CFG5 = dict(H=1080, W=1920, RAYS=8192, SPP=32, VIEW=0, TRIS=1_000_000, SLF_H=256, SEED=0, SCENE_SEED=1)


def cfg5_rays_pick():
    """the 8 192 pixels of the fixture (and of tests/test_cfg5_full_size.py): torch.randint under a CPU generator seeded 0"""
    import torch
    g = torch.Generator(device="cpu").manual_seed(0)
    return torch.randint(0, CFG5["H"] * CFG5["W"], (CFG5["RAYS"],), generator=g)


def cfg5_draw(oracle, k, shape):
    """what the k-th torch.rand(shape) of the replayed path_tracing_single call returned: the first numel values of Philox stream 16 + k"""
    n = int(np.prod(shape))
    return np.ascontiguousarray(oracle.philox_u2(CFG5["SEED"], 0, 16 + k, (n + 1) // 2).reshape(-1)[:n].reshape(shape), np.float32)


def flipped_pixels(L, L_ref, tol=1e-4):
    """pixels whose radiance differs from the reference's by more than rounding in some channel: a path of the pixel took another discrete turn"""
    L = np.asarray(L, np.float64); L_ref = np.asarray(L_ref, np.float64)
    return (np.abs(L - L_ref) > tol * np.maximum(np.abs(L_ref), 1e-3)).any(-1)
