"""The multi-GPU sampling path.  CPU part: ``bench.py --gpus N`` really starts N ranks (as a child torchrun, before it
touches any GPU).  GPU part (skipped on boxes with fewer than two GPUs): the product sampler under
``distributed.sample_sharded`` on RCCL (backend "nccl"), one process per GPU, bit-identical to single-rank samples."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT


def test_bench_gpus_flag_launches_n_ranks():
    env = dict(os.environ, PRD_BENCH_DRY_LAUNCH="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 0, out.stderr.decode()
    cmd = json.loads(out.stdout.decode().strip().splitlines()[-1])["launch"]
    assert "--nproc-per-node=4" in cmd and "torch.distributed.run" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")


@pytest.mark.gpu
def test_sharded_sampling_on_rccl_matches_single_rank():
    n = torch.cuda.device_count()                 # counting devices does not initialise the GPU in this process
    if n < 2:
        pytest.skip("needs at least two GPUs (the driver's multi-GPU tier)")
    world = min(n, 4)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "tests", "_nccl_worker.py")],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0, out.stdout.decode()[-2000:]
