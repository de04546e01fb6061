"""bench.py's output contract (the driver parses exactly one JSON line): run it as the driver does, on a small workload, and check the
fields, their types and their internal consistency.  The default-size run is the driver's job; this guards the format."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _run(*extra):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--height", "120", "--width", "160", "--spp", "32", "--tris", "20000",
           "--slf-res", "64", "--views", "4", "--cpu-seconds", "0.5", "--no-extras"] + list(extra)
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly ONE line on stdout:\n" + r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_single_gpu_line():
    d = _run()
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    rays = 120 * 160 * 32 * 7
    assert d["config"]["rays_per_step"] <= rays and d["config"]["rays_per_step"] > 0.5 * rays
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) <= 0.02 * d["value"]        # value == rays / time
    rf = d["roofline"]
    assert rf["launches"] == 3 and rf["launch_ms"] > 0 and rf["work_per_ray"]["nodes_per_ray"] > 1
    # the committed PMC profile is for the 1080p launch: it must be REFUSED for this one, and the fraction then left open rather than invented
    assert rf["pmc_source"] != "committed" and rf["bound"] is None and rf["frac"] is None and rf["traffic"] is None
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mrays/s" and "sample" in cb
    assert d["multi_gpu"]["rccl_ranks_seen"] == 1 and d["multi_gpu"]["gather_ms"] is None


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_line():
    """--gpus 2 as a plain command starts its own two workers; with IRIS_BENCH_BACKEND=gloo they may share the one GPU of this box
    (a functional check of the N > 1 control flow: sharded bake, overlapped gather, cross-rank image check -- never a measurement)."""
    env = dict(os.environ, IRIS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "120", "--width", "160", "--spp", "32",
           "--tris", "20000", "--slf-res", "64", "--views", "4", "--cpu-seconds", "0", "--no-extras", "--no-roofline"]
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    mg = d["multi_gpu"]
    assert mg["rccl_ranks_seen"] == 2 and len(mg["per_rank_ms_per_step"]) == 2 and mg["gather_ms"] is not None
    assert mg["gather_overlapped"] is True and mg["gathered_image_identical_on_all_ranks"] is True
