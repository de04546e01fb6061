"""Oracle (CPU baseline) thread scaling on the current host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, oracle
from tools import synth
r = synth.room(1, 1_000_000)
sc = oracle.Scene(r["vertices"], r["faces"])
s = synth.slf_for(r["vertices"], r["faces"], 256); e = synth.emitters_for(r["vertices"], r["faces"], r["is_emitter"])
slf = oracle.VoxelSLF(s["inds"], s["radiance"], s["voxel_min"], s["voxel_max"]); em = oracle.SLFEmitter(e["is_emitter"], e["emitter_radiance"], e["emitter_area"], slf)
K, c2w = synth.camera(270, 480, 0)
o, d = oracle.raygen_real(K, c2w, 270, 480)
p, n, uv, idx, valid = sc.ray_intersect(o, d)
pos, nrm = p[valid], n[valid]
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (1, 8, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    oracle.set_num_threads(th)
    P = min(len(pos), max(2000, th * 400))
    t = time.time(); oracle.bake(sc, em, pos[:P], nrm[:P], 128, seed=1); dt = time.time() - t
    print("threads %3d: %7.2f Mrays/s (%d rays, %.2f s)" % (th, P * 128 / dt / 1e6, P * 128, dt), flush=True)
