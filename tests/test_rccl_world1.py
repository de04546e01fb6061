"""RCCL executed once on the one GPU a test box has: a process group of world size 1 on the "nccl" backend (= RCCL on ROCm), and a view sent
through MapGatherer's COLLECTIVE branch (force_collective) in both modes -- dist.gather with a list of receive views, all_gather_into_tensor,
and iris_unstripe_maps reading a buffer RCCL wrote.  At world 1 the gathered image must equal the local maps bit for bit (one rank owns
every stripe).  The multi-rank arithmetic of the same code is covered over gloo (tests/test_sharding.py, test_sharding_gpu.py); what this adds
is that librccl loads, creates a communicator on the device and moves the 13-map buffers of a 1080p view.

The process group lives in a child process (spawn): a test session must not keep a NCCL communicator, and a hang would otherwise take the
session with it."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO, free_port

pytestmark = pytest.mark.gpu


def _worker(port, q):
    import datetime
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    try:
        from iris_amd import sharding as sh
        assert dist.get_backend() == "nccl"
        out = {}
        for (H, W) in ((40, 56), (1080, 1920)):                       # a small view and BASELINE configs[3]'s 13 x 1080p maps (323 MB)
            g = torch.Generator(device="cpu").manual_seed(H)
            local = torch.rand(13, H * W, 3, generator=g).to(dev)
            assert sh.local_pixel_ids(H, W, 1, 0).numel() == H * W
            for mode in ("gather", "all_gather"):
                ga = sh.MapGatherer(H, W, 1, 0, 13, dev, mode=mode, force_collective=True)
                assert ga.collective and ga.receives and ga.recv is not None
                for k in range(2):                                    # buffers reused over views
                    full = ga(local * (k + 1))
                    torch.cuda.synchronize()
                    assert full is ga.full and full.data_ptr() != local.data_ptr()
                    assert torch.equal(full, local * (k + 1)), (H, W, mode, k)
                out[f"{H}x{W}:{mode}"] = True
                del ga
            # the one-shot form
            assert torch.equal(sh.gather_maps(local, H, W, 1, 0, force_collective=True), local)
            # without force_collective a world of one is the identity (no buffers, no collective)
            assert sh.gather_maps(local, H, W, 1, 0) is local
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        assert float(t.item()) == 1.0
        q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_process_group_of_one_runs_the_collective_branch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(free_port(), q))
    p.start()
    out = q.get(timeout=500)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert out == {"40x56:gather": True, "40x56:all_gather": True, "1080x1920:gather": True, "1080x1920:all_gather": True}


def test_device_maps_must_be_float32():
    from iris_amd import sharding as sh
    with pytest.raises(ValueError):
        sh.MapGatherer(8, 8, 2, 0, 1, torch.device("cuda", 0), dtype=torch.float16)
