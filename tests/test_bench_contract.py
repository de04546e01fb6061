"""bench.py's output contract (the driver parses exactly one JSON line): run it as the driver does, on a small workload, and check the
fields, their types and their internal consistency.  The default-size run is the driver's job; this guards the format."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _run(*extra, env_extra=None, extras=True):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--height", "120", "--width", "160", "--spp", "32", "--tris", "20000",
           "--slf-res", "64", "--views", "4", "--cpu-seconds", "0.5"] + ([] if extras else ["--no-extras"]) + list(extra)
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env_extra or {})))
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
    assert cb["threads"] == cb["cores"] and cb["repeats"] == 3 and len(cb["seconds_by_repeat"]) == 3 and "cgroup_cpus" in cb
    mg = d["multi_gpu"]                      # no process group at N = 1: nothing about a backend is claimed
    assert mg["backend"] is None and mg["process_group"] is False and mg["ranks_seen_by_all_reduce"] is None and mg["gather_ms"] is None
    pc = d["parity_check"]                   # the timed maps themselves, against the device-arithmetic oracle
    assert pc["bit_exact"] is True and pc["maps"] == 13 and pc["pixels"] >= 4096 and pc["mismatches"] == []
    ex = d["extras"]
    assert ex["n_lobes_1_diffuse_only"]["mrays_per_s"] > 0 and ex["reference_spp_mix"]["spp"] == [256, 64, 128, 128, 128, 128, 128]


@pytest.mark.timeout(900)
def test_forced_process_group_of_one_runs_rccl():
    """IRIS_BENCH_FORCE_PG=1: RCCL's communicator, the gather of a view and the permutation of the received buffer execute on this one GPU"""
    for collective in ("gather", "all_gather"):
        d = _run("--gather", collective, env_extra={"IRIS_BENCH_FORCE_PG": "1"}, extras=False)
        mg = d["multi_gpu"]
        assert mg["backend"] == "rccl" and mg["process_group"] is True and mg["forced_at_world_1"] is True and mg["ranks_seen_by_all_reduce"] == 1
        assert mg["collective"] == collective and mg["gather_ms"] is not None and mg["gathered_image_matches_what_the_ranks_sent"] is True
        assert d["parity_check"]["bit_exact"] is True


def _two_ranks(*extra, env_extra=None, expect_ok=True, n=2):
    """--gpus N as a plain command starts its own N workers; with IRIS_BENCH_BACKEND=gloo they may share the one GPU of this box
    (a functional check of the N > 1 control flow -- never a measurement)."""
    import time
    env = dict(os.environ, IRIS_BENCH_BACKEND="gloo", **(env_extra or {}))
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--height", "120", "--width", "160", "--spp", "32",
           "--tris", "20000", "--slf-res", "64", "--views", "4", "--cpu-seconds", "0", "--no-extras", "--no-roofline"] + list(extra)
    t0 = time.time()
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600, env=env)
    dt = time.time() - t0
    if not expect_ok:
        return r, dt
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), dt


@pytest.mark.timeout(900)
@pytest.mark.parametrize("collective", ["gather", "all_gather"])
def test_two_ranks_on_one_gpu_line(collective):
    """sharded bake, overlapped gather (to rank 0: north_star's single gather; or to every rank), check of the gathered image against what was sent"""
    d, _ = _two_ranks("--gather", collective)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    mg = d["multi_gpu"]
    assert mg["ranks_seen_by_all_reduce"] == 2 and mg["backend"] == "gloo" and len(mg["per_rank_ms_per_step"]) == 2 and mg["gather_ms"] is not None
    assert mg["gather_overlapped"] is True and mg["collective"] == collective and mg["gathered_image_matches_what_the_ranks_sent"] is True


@pytest.mark.timeout(900)
def test_the_drivers_launch_form():
    """the exact form the driver uses for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from torchrun's environment) -- here with two gloo ranks sharing the GPU: ONE JSON line on stdout, from rank 0"""
    from conftest import free_port
    env = dict(os.environ, IRIS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "120", "--width", "160", "--spp", "32", "--tris", "20000", "--slf-res", "64",
           "--views", "4", "--cpu-seconds", "0", "--no-extras", "--no-roofline"]
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["multi_gpu"]["ranks_seen_by_all_reduce"] == 2
    assert d["multi_gpu"]["gathered_image_matches_what_the_ranks_sent"] is True and d["parity_check"]["bit_exact"] is True


@pytest.mark.timeout(900)
def test_eight_ranks_on_one_gpu_line():
    """BASELINE configs[3]'s rank count: `bench.py --gpus 8` starts eight workers (spawn_workers), they rendezvous, pass the barriers and the timed loop, send 15
    stripes of 8 rows through the gather (seven ranks own two, the eighth one: padded send rows) and rank 0 prints the one line -- eight gloo ranks sharing this
    box's GPU: functional evidence only, never a measurement."""
    d, _ = _two_ranks("--gather", "gather", n=8)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    mg = d["multi_gpu"]
    assert mg["ranks_seen_by_all_reduce"] == 8 and mg["backend"] == "gloo" and len(mg["per_rank_ms_per_step"]) == 8
    assert mg["collective"] == "gather" and mg["gathered_image_matches_what_the_ranks_sent"] is True
    rays = 120 * 160 * 32 * 7
    assert 0.5 * rays < d["config"]["rays_per_step"] <= rays                      # strong scaling: ONE view per step, whatever N


@pytest.mark.timeout(1500)
def test_eight_ranks_full_size_line_is_complete():
    """BASELINE configs[3] as the driver will run it -- `bench.py --gpus 8`, 1920 x 1080, SPP 128, the 1.0 M-triangle room -- with the eight ranks sharing this box's
    one GPU over gloo (functional, never a measurement): the line rank 0 prints must be as complete as the N = 1 line -- `roofline` with a fraction (the counters of
    rank 0's stripes: profiles/pmc_r6_world8.json, taken with --emulate-world 8 and labelled so; a profile of other kernel sources is refused with the reason),
    `cpu_baseline` (run by rank 0 after the timed region), the parity check of the timed maps, the gathered image equal to what the ranks sent."""
    import time
    env = dict(os.environ, IRIS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "3", "--cpu-repeats", "1", "--no-extras"]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=1400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["rays_per_step"] == 1920 * 1080 * 128 * 7
    mg = d["multi_gpu"]
    assert mg["ranks_seen_by_all_reduce"] == 8 and mg["gathered_image_matches_what_the_ranks_sent"] is True and mg["collective"] == "gather"
    assert d["parity_check"]["bit_exact"] is True
    rl = d["roofline"]
    assert rl["rays_per_launch"] == 17 * 8 * 1920 * 128 * 7                      # rank 0 owns 17 of the 135 stripes of 8 rows
    if rl["frac"] is None:
        assert "stale" in rl["pmc_source"], rl["pmc_source"]                      # the committed counters are of other kernel sources: refused, with the reason
    else:
        assert rl["pmc_file"] == "profiles/pmc_r6_world8.json" and "emulate-world 8" in rl["pmc_source"] and 0.0 < rl["frac"] <= 1.0 and rl["traffic"] > 0      # (eight ranks share ONE GPU here: the fraction follows this run's contended ray rate and means nothing as a measurement)
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and d["gpu_over_cpu"] > 0
    print("eight ranks, full size: %.0f s" % (time.time() - t0), rl["frac"], rl["pmc_source"], cb["value"])


@pytest.mark.timeout(900)
def test_two_ranks_sharded_by_views():
    """SURVEY 8(e)'s fallback for the view sequence: whole views per rank, no collective, weak scaling"""
    d, _ = _two_ranks("--shard", "views")
    one, _ = _two_ranks("--shard", "views", "--steps", "1")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "no collective" in d["config"]["sharding"]
    assert d["multi_gpu"]["gather_ms"] is None and d["multi_gpu"]["collective"] is None
    rays_view = 120 * 160 * 32 * 7
    assert 0.5 * 2 * rays_view < d["config"]["rays_per_step"] <= 2 * rays_view                # two whole views per step
    assert one["config"]["rays_per_step"] == pytest.approx(d["config"]["rays_per_step"], rel=0.05)


@pytest.mark.timeout(600)
def test_a_rank_dying_mid_run_ends_the_job_quickly():
    """rank 1 raises in its second step while rank 0 waits in the gather: bench.py --gpus 2 must exit non-zero within seconds, not hang"""
    r, dt = _two_ranks(env_extra={"IRIS_BENCH_FAIL_RANK": "1", "IRIS_BENCH_FAIL_STEP": "1"}, expect_ok=False)
    assert r.returncode != 0, r.stdout[-500:]
    assert "IRIS_BENCH_FAIL_RANK" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]                 # no result line from a failed job
    ok, dt_ok = _two_ranks()
    assert dt < dt_ok + 30.0, (dt, dt_ok)                                                       # ended about as fast as a good run ends

