"""Data-parallel equivalence on real devices: runs only where >= 2 GPUs are visible (the round's 1-GPU box skips it,
an 8-GPU node runs it).  Two fresh ranks are started as child processes (never an exec from this GPU-initialised
process); what they assert is in tests/_dp_gpu_child.py."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_children(env):
    child = os.path.join(ROOT, "tests", "_dp_gpu_child.py")
    procs = [subprocess.Popen([sys.executable, child, ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"DP_ALL_OK {r}" in o, o[-4000:]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_ranks_equal_one_rank_and_stay_in_sync():
    """Two GPUs, RCCL."""
    _run_children(dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0"))


def test_two_ranks_equal_one_rank_and_stay_in_sync_on_one_gpu():
    """The same assertions with both ranks on cuda:0 and gloo collectives: runs on the 1-GPU box, so the data-parallel
    equivalence (2 x bs 2 == 1 x bs 4 gradients, bit-identical parameters after graph-replayed steps, BatchNorm buffer
    semantics) is checked on real device tensors every round, not only where two GPUs are visible."""
    _run_children(dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29573", WORLD_SIZE="2", SHINEON_DIST_BACKEND="gloo",
                       SHINEON_LOCAL_DEVICE="0"))


@pytest.mark.parametrize("extra", [[], ["--no-pipeline"], ["--config", "c3"]])
def test_bench_multi_rank_code_path_two_ranks_on_one_gpu(extra, tmp_path):
    """The N > 1 path of bench.py / trainer.ChainedTrainStep (flat parameter + buffer broadcasts, schedule choice from the
    measured exchange time, asynchronous gradient exchange on both streams, MAX-over-ranks timing, rank-0 JSON line) run
    for real with world_size = 2 - both ranks on cuda:0, collectives over gloo because RCCL refuses two ranks per device.
    Functional coverage only (the timing means nothing); the 8-GPU scaling run itself belongs to the driver."""
    import json

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", WORLD_SIZE="2", SHINEON_DIST_BACKEND="gloo",
               SHINEON_LOCAL_DEVICE="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-hbm-table"] + extra
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{e[-3000:]}"
    line = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["value"] > 0 and line["scaling"] == "weak"
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]   # only rank 0 prints the JSON line
