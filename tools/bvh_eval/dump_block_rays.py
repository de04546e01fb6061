"""Dump PIXEL-BLOCK bake rays for tools/bvh_eval/wavesim (round 6: ray reordering across tiles, decided on the kernel's own rays).

    python tools/bvh_eval/dump_block_rays.py /tmp/block_rays.bin [block=64] [n_blocks=3] [lobes=0,2,4,6] [seed=1] [tris=1000000] [spp=128]

The bench workload's rays as the timed kernel generates them (view 0 of the circle, 1920 x 1080, the kernels' Philox stream keyed by image pixel and lobe, the oracle's
samplers in device-arithmetic mode, origin = position + RayEpsilon * wi), for square blocks of `block` x `block` image pixels spread over the image.  Unlike dump_rays.py
nothing is sorted here: rays are written pixel-major (y, x, sample) and the simulator forms tiles / windows and orders them itself.
File: int64 n_groups, int64 block, int64 spp; per group int32 lobe, int32 x0, int32 y0, then block * block * spp x (o.xyz, d.xyz) float32.  Uses the CPU oracle (test infrastructure)."""
import os
import struct
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tools import synth      # noqa: E402
import oracle                # noqa: E402

H, W = 1080, 1920
RAY_EPS = np.float32(8.940696716308594e-05)


def main():
    out = sys.argv[1]
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n_blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lobes = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,2,4,6").split(",")]
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    tris = int(sys.argv[6]) if len(sys.argv) > 6 else 1_000_000
    spp = int(sys.argv[7]) if len(sys.argv) > 7 else 128
    oracle.build()
    room = synth.room(seed, tris)
    osc = oracle.Scene(room["vertices"], room["faces"])
    K, c2w = synth.camera(H, W, 0)
    xs, ds = oracle.raygen_real(K, c2w, H, W)
    rough = np.linspace(0.02, 1.0, 6, dtype=np.float32)
    oracle.set_mode(1)
    with open(out, "wb") as fh:
        fh.write(struct.pack("<qqq", n_blocks * len(lobes), block, spp))
        for b in range(n_blocks):
            # blocks spread over the image (aligned to `block`, as the host's block-ordered pixel list cuts them)
            y0 = (((b * 2 + 1) * H) // (2 * n_blocks)) // block * block
            x0 = ((b * 577 + 300) % (W - block)) // block * block
            yy, xx = np.meshgrid(np.arange(y0, y0 + block), np.arange(x0, x0 + block), indexing="ij")
            pix = (yy * W + xx).reshape(-1)
            pos, nrm, _, idx, valid = osc.ray_intersect(xs[pix], ds[pix])
            assert valid.all()
            wo = -ds[pix]
            for lobe in lobes:
                u = np.concatenate([oracle.philox_u2(0, int(i0), lobe, spp) for i0 in pix.astype(np.int64) * spp])
                n_rep = np.repeat(nrm, spp, 0); w_rep = np.repeat(wo, spp, 0); p_rep = np.repeat(pos, spp, 0)
                if lobe == 0:
                    wi, _, _ = oracle.sample_diffuse(u, n_rep)
                else:
                    wi, _, _, _ = oracle.sample_specular(u, w_rep, n_rep, rough[lobe - 1])
                o = (p_rep + RAY_EPS * wi).astype(np.float32)
                fh.write(struct.pack("<iii", lobe, x0, y0))
                fh.write(np.ascontiguousarray(np.concatenate([o, wi], 1), np.float32).tobytes())
                print("block", b, (x0, y0), "lobe", lobe, flush=True)
    oracle.set_mode(0)


if __name__ == "__main__":
    main()
