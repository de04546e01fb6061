"""The host harness behind EXPERIMENTS.md round 5 (tools/bvh_eval): kept alive by the CPU suite, because its counts are evidence.

  * bvh_eval builds, traces the bake kernel's own tile-structured rays (tools/bvh_eval/dump_rays.py: oracle samplers, Philox stream, LDS-sort order) on a small room and
    reproduces the QUALITATIVE findings the round's decisions rest on: an 8-wide collapse visits fewer nodes but tests more boxes per ray; children in the ray octant's
    split-axis order cost (almost) no visits against sorting by entry distance; a hit-distance predictor from the list neighbour rarely hits on the diffuse lobe
  * the static instruction count of one BVH4 / BVH8 node visit (tools/bvh_eval/node_step8_isa.sh, hipcc cross-compile with the shipped flags)"""
import json
import os
import shutil
import subprocess
import sys

import pytest

from conftest import REPO

BVH_EVAL = os.path.join(REPO, "tools", "bvh_eval")


@pytest.fixture(scope="module")
def harness(tmp_path_factory, oracle_mod):
    tmp = tmp_path_factory.mktemp("bvh_eval")
    exe = str(tmp / "bvh_eval")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(REPO, "iris_amd", "csrc"), os.path.join(BVH_EVAL, "bvh_eval.cpp"),
                    os.path.join(REPO, "iris_amd", "csrc", "bvh_build.cpp"), "-lpthread", "-o", exe], check=True, timeout=600)
    room, rays = str(tmp / "room.bin"), str(tmp / "rays.bin")
    subprocess.run([sys.executable, os.path.join(BVH_EVAL, "dump_room.py"), room, "0", "60000"], check=True, timeout=600, cwd=REPO)
    subprocess.run([sys.executable, os.path.join(BVH_EVAL, "dump_rays.py"), rays, "3", "0", "60000"], check=True, timeout=600, cwd=REPO)

    def run(**env):
        out = subprocess.run([exe, room, "0", "4", "1", "0.7"], check=True, capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, BVH_EVAL_RAYS=rays, **{k: str(v) for k, v in env.items()})).stdout
        return json.loads(out)
    return run


@pytest.mark.timeout(900)
def test_harness_counts(harness):
    b4 = harness(BVH_EVAL_WIDTH=4, BVH_EVAL_ORDER="octant")
    b4d = harness(BVH_EVAL_WIDTH=4, BVH_EVAL_ORDER="distance")
    b8 = harness(BVH_EVAL_WIDTH=8, BVH_EVAL_ORDER="octant")
    assert b4["rays"] == 3 * 7 * 4096 and b4["hit_frac"] == 1.0                     # a closed room: every ray hits
    assert 10 < b4["nodes_per_ray"] < 40 and 2 < b4["tris_per_ray"] < 12
    assert abs(b4["nodes_per_ray"] - b4d["nodes_per_ray"]) <= 0.03 * b4d["nodes_per_ray"]      # the split-axis order costs (almost) no visits
    # the wide node: fewer visits, MORE box tests
    assert b8["nodes"] < 0.6 * b4["nodes"] and b8["nodes_per_ray"] < 0.8 * b4["nodes_per_ray"]
    assert b8["slab_tests_per_ray"] > 1.2 * b4["slab_tests_per_ray"]
    # the predictor: the list neighbour's triangle is hit by a few per cent of the diffuse rays; the mirror-like lobe is the exception
    p = harness(BVH_EVAL_WIDTH=4, BVH_EVAL_ORDER="octant", BVH_EVAL_PRED=1)
    by_lobe = {r["lobe"]: r for r in p["per_lobe"]}
    assert by_lobe[0]["predictor_hit_rate"] < 0.05 < 0.3 < by_lobe[1]["predictor_hit_rate"]
    assert by_lobe[0]["nodes_per_ray_with_predictor"] <= by_lobe[0]["nodes_per_ray"]           # a valid bound never adds visits
    assert all(r["tris_per_ray_with_predictor"] > r["tris_per_ray"] for r in p["per_lobe"])    # ... but costs its triangle test


@pytest.mark.skipif(not shutil.which("/opt/rocm/bin/hipcc"), reason="hipcc not available")
@pytest.mark.timeout(900)
def test_node_visit_instruction_counts():
    out = subprocess.run([os.path.join(BVH_EVAL, "node_step8_isa.sh")], check=True, capture_output=True, text=True, timeout=800).stdout
    d = json.loads(out.strip().splitlines()[-1])
    b4, b8 = d["bvh4"], d["bvh8"]
    assert b4["vector_loads"] == 4 and b4["v_fma_mix_f32"] == 24 and b4["scratch"] == 0 and 80 <= b4["vector_alu"] <= 100
    assert b8["vector_loads"] == 7 and b8["v_fma_mix_f32"] == 48 and b8["scratch"] == 0
    assert b8["vector_alu"] > 1.7 * b4["vector_alu"]                                          # what decided against the per-lane BVH8: 16.6 x 176 > 24.2 x 91
    assert b8["vgprs_of_the_bare_loop"] > b4["vgprs_of_the_bare_loop"] + 15
