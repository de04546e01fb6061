"""Sanitizer runs on the CPU builds (GPU AddressSanitizer is not available on the pool): the oracle compiled with
-fsanitize=address,undefined replays golden-vector workloads, and the host BVH builder of libiris_hip.so is compiled standalone
with the same flags and checked for its structural invariants on random / degenerate inputs."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

ASAN = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
pytestmark = pytest.mark.skipif(not os.path.isabs(ASAN) or not os.path.exists(ASAN), reason="libasan not installed")

SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import oracle
from conftest import golden
g = golden("bake_box.npz")
sc = oracle.Scene(g["verts"], g["faces"])
slf = oracle.VoxelSLF(g["slf_inds"], g["slf_radiance"], float(g["voxel_min"]), float(g["voxel_max"]))
em = oracle.SLFEmitter(g["is_emitter"], g["emitter_radiance"], g["emitter_area"], slf)
rng = np.random.default_rng(0)
o = rng.random((2000, 3)).astype(np.float32) * [4, 3, 2.6]; d = rng.standard_normal((2000, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
pos, nrm, uv, idx, valid = sc.ray_intersect(o, d)
assert valid.all()
for mode in (False, True):
    ctx = oracle.device_arithmetic() if mode else __import__("contextlib").nullcontext()
    with ctx:
        (Ld,) = oracle.bake(sc, em, pos[:300], nrm[:300], 16, seed=1, stream=0)
        a, b = oracle.bake(sc, em, pos[:300], nrm[:300], 16, wo=-d[:300], roughness=np.float32(0.216), seed=1, stream=2)
assert np.isfinite(Ld).all() and np.isfinite(a).all() and np.isfinite(b).all()
(Le,) = oracle.bake(sc, em, pos[:0], nrm[:0], 16, seed=1, stream=0)          # empty input
img = rng.random((37, 53, 3)).astype(np.float32)
val = rng.random((37, 53)) > 0.1
out = oracle.denoise(img, nrm[:37 * 53].reshape(37, 53, 3) if len(nrm) >= 37 * 53 else None, None, val)
out = oracle.denoise(img, None, None, None, iterations=2)
s = golden("shade_cached.npz")
rows = oracle.cache_pack(s["map_0"], [s[f"map_{1 + j}"] for j in range(6)], [s[f"map_{7 + j}"] for j in range(6)])
L, ga, gm, gr = oracle.shade_cached(rows, s["idx"], s["albedo"], s["metallic"], s["roughness"], s["gL"])
assert np.array_equal(L, s["L"])
u = oracle.philox_u2(3, 0, 1, 1000) if hasattr(oracle, "philox_u2") else None
print("ASAN-ORACLE-OK")
"""


def test_oracle_under_asan_ubsan(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "asan"])
    env = dict(os.environ, REPO=REPO, IRIS_ORACLE_LIB=os.path.join(REPO, "oracle", "libiris_oracle_asan.so"), LD_PRELOAD=ASAN,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ASAN-ORACLE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_bvh_builder_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "bvh_build_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", exe,
                           os.path.join(REPO, "tests", "native", "bvh_build_check.cpp"), os.path.join(REPO, "iris_amd", "csrc", "bvh_build.cpp"), "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "all BVH builder checks passed" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr, r.stderr[-3000:]
