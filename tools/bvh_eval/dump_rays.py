"""Dump TILE-STRUCTURED bake rays for tools/bvh_eval/bvh_eval (round 5: hit-distance seeding with a real predictor, wide-node studies on the kernel's own rays).

    python tools/bvh_eval/dump_rays.py /tmp/rays.bin [tiles=24] [seed=1] [tris=1000000]

The bench workload's rays as the timed kernel sees them: a 1920 x 1080 view of the bench scene, tiles of 32 consecutive valid pixels x SPP 128 = 4096 rays
(iris_bake.h tile_body), for each of the 7 lobes; per ray the Philox uniforms the kernels draw (seed 0, image_pixel * spp + sample, lobe), the oracle's
samplers in device-arithmetic mode, origin = position + RayEpsilon * wi, and inside a tile the order of the LDS counting sort: by direction bin
(iris_tile.h dir_bin: octant | 8 x 4 cells; the order inside a bin is the arrival order of an atomic on the GPU -- ray-index order here).
File: int64 n_groups, int64 rays_per_group, then per group int32 lobe + rays_per_group x (o.xyz, d.xyz) float32.  Uses the CPU oracle (test infrastructure)."""
import os
import struct
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tools import synth      # noqa: E402
import oracle                # noqa: E402

H, W, SPP, TILE_PX = 1080, 1920, 128, 32
RAY_EPS = np.float32(8.940696716308594e-05)


def dir_bin(d):
    """iris_tile.h dir_bin (the reciprocals there are 1-ulp approximations: bins on a cell border may differ, which a harness does not care about)"""
    ax, ay, az = np.abs(d[:, 0]), np.abs(d[:, 1]), np.abs(d[:, 2])
    inv = 1.0 / (ax + ay + az + 1e-30)
    a, b = ax * inv, ay * inv
    v = b / (1.0 - a + 1e-30)
    iu = np.minimum(7, (a * 8.0).astype(np.int32)); iv0 = np.minimum(3, (v * 4.0).astype(np.int32))
    iv = np.where(iu & 1, 3 - iv0, iv0)
    octant = (d[:, 0] < 0).astype(np.int32) | ((d[:, 1] < 0).astype(np.int32) << 1) | ((d[:, 2] < 0).astype(np.int32) << 2)
    return (octant << 5) | (iu << 2) | iv


def main():
    out = sys.argv[1]
    n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    tris = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
    oracle.build()
    room = synth.room(seed, tris)
    osc = oracle.Scene(room["vertices"], room["faces"])
    K, c2w = synth.camera(H, W, 0)
    xs, ds = oracle.raygen_real(K, c2w, H, W)
    rough = np.linspace(0.02, 1.0, 6, dtype=np.float32)
    groups = []
    oracle.set_mode(1)
    for t in range(n_tiles):
        # a tile = 32 consecutive pixels (the closed room: every pixel is valid), spread evenly over the image, never across a row end
        row = (t * H) // n_tiles + H // (2 * n_tiles); col = ((t * 577) % (W - TILE_PX))
        pix = row * W + col + np.arange(TILE_PX)
        pos, nrm, _, idx, valid = osc.ray_intersect(xs[pix], ds[pix])
        assert valid.all()
        wo = -ds[pix]
        for lobe in range(7):
            idx0 = (pix[:, None].astype(np.int64) * SPP + np.arange(SPP)[None, :]).reshape(-1)
            u = np.concatenate([oracle.philox_u2(0, int(i0), lobe, SPP) for i0 in pix.astype(np.int64) * SPP])     # (32 * 128, 2): pixel-major, as the tile's ray index r = pl * spp + s
            n_rep = np.repeat(nrm, SPP, 0); w_rep = np.repeat(wo, SPP, 0); p_rep = np.repeat(pos, SPP, 0)
            if lobe == 0:
                wi, _, _ = oracle.sample_diffuse(u, n_rep)
            else:
                wi, _, _, _ = oracle.sample_specular(u, w_rep, n_rep, rough[lobe - 1])
            o = (p_rep + RAY_EPS * wi).astype(np.float32)
            order = np.argsort(dir_bin(wi), kind="stable")
            groups.append((lobe, np.concatenate([o[order], wi[order]], 1).astype(np.float32)))
    oracle.set_mode(0)
    with open(out, "wb") as fh:
        fh.write(struct.pack("<qq", len(groups), TILE_PX * SPP))
        for lobe, rays in groups:
            fh.write(struct.pack("<i", lobe)); fh.write(np.ascontiguousarray(rays).tobytes())
    print(out, len(groups), "groups of", TILE_PX * SPP, "rays")


if __name__ == "__main__":
    main()
