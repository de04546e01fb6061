"""Dump the synthetic room mesh for tools/bvh_eval/bvh_eval:  python tools/bvh_eval/dump_room.py /tmp/room.bin [seed] [tris]"""
import os, struct, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import synth
out = sys.argv[1]; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1; tris = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
r = synth.room(seed, tris)
v = np.ascontiguousarray(r["vertices"], np.float32); f = np.ascontiguousarray(r["faces"], np.int32)
with open(out, "wb") as fh:
    fh.write(struct.pack("<qq", v.shape[0], f.shape[0])); fh.write(v.tobytes()); fh.write(f.tobytes())
print(out, v.shape, f.shape)
