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


def _torchrun_one_rank(script_args, port, timeout=900):
    """A FRESH child process (never an exec of the pytest process, which may have initialised the GPU): one rank under
    torch.distributed.run, backend "nccl" = RCCL."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, (out.stdout.decode()[-1500:], out.stderr.decode()[-1500:])
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert lines, out.stdout.decode()[-1500:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_bench_under_one_rank_rccl_launcher():
    """bench.py exactly as the driver launches it for N > 1, with one rank: RCCL init, the samples' all_gather inside the timed
    region, and the shard check with a batch of two complexes of the bench shape (N = 320) per GPU."""
    line = _torchrun_one_rank([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--samples-per-gpu", "2",
                               "--no-cpu-baseline", "--no-traffic"], 29621)
    assert line["n_gpus"] == 1 and line["config"]["backend"] == "nccl (RCCL)"
    chk = line["config"]["sharded_sample_check"]
    assert chk["identical_to_same_batch_size_recompute"] is True and chk["batch_size"] == 2 and chk["rel_l2_vs_sample_drawn_alone"] < 2e-5
    assert line["config"]["outputs_finite"] is True and line["value"] > 0


@pytest.mark.gpu
def test_training_step_under_one_rank_rccl_launcher():
    """tools/train_bench.py under the same launcher: the flat gradient all-reduce of training.all_reduce_gradients runs on RCCL."""
    line = _torchrun_one_rank([os.path.join(ROOT, "tools", "train_bench.py"), "--steps", "2", "--warmup", "1"], 29622)
    assert line["backend"].startswith("nccl (RCCL)") and line["n_gpus"] == 1
    assert all(x == x and abs(x) < 1e6 for x in line["losses"])
