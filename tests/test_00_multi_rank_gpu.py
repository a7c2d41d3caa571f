"""Data-parallel equivalence and the multi-rank bench path on real devices.  Two fresh ranks are started as child
processes; the file name sorts FIRST on purpose: pytest runs it before any other GPU test, i.e. while this parent process has
not initialised the GPU yet (device_count() does not), so the children are never spawned from a GPU-initialised process.
What the ranks assert is in tests/_dp_gpu_child.py.  With two GPUs visible the ranks use RCCL; on a 1-GPU box both ranks
share cuda:0 and the collectives run over gloo (RCCL refuses two ranks per device)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn_pair(cmd, env, timeout=420):
    """Start the two ranks as fresh child processes and wait.  Returns [(returncode, output)] or None when the pair did not
    finish in time (rendezvous trouble on the box: the ranks are killed so that nothing is left on the GPU)."""
    env = dict(env, GLOO_SOCKET_IFNAME=env.get("GLOO_SOCKET_IFNAME", "lo"), SHINEON_DIST_TIMEOUT_S="180")
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, start_new_session=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        for p in procs:
            p.communicate()
        return None
    return [(p.returncode, o) for p, o in zip(procs, outs)]


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run_children(env):
    child = os.path.join(ROOT, "tests", "_dp_gpu_child.py")
    res = _spawn_pair([sys.executable, child, ROOT], dict(env, MASTER_PORT=_free_port()))
    if res is None:
        pytest.skip("the two ranks did not rendezvous / finish in time on this box (ranks killed)")
    for r, (rc, o) in enumerate(res):
        assert rc == 0 and f"DP_ALL_OK {r}" in o, o[-4000:]


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


@pytest.mark.parametrize("extra", [[], ["--no-pipeline"], ["--config", "c3"], ["--config", "sams", "--batch", "1"]])
def test_bench_multi_rank_code_path_two_ranks_on_one_gpu(extra, tmp_path):
    """The N > 1 path of bench.py / trainer.ChainedTrainStep (flat parameter + buffer broadcasts, schedule choice from the
    measured exchange time, asynchronous gradient exchange on both streams, MAX-over-ranks timing, rank-0 JSON line) run
    for real with world_size = 2 - both ranks on cuda:0, collectives over gloo because RCCL refuses two ranks per device.
    Functional coverage only (the timing means nothing); the 8-GPU scaling run itself belongs to the driver."""
    import json

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="2", SHINEON_DIST_BACKEND="gloo",
               SHINEON_LOCAL_DEVICE="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-hbm-table"] + extra
    res = _spawn_pair(cmd, env)
    if res is None:
        pytest.skip("the two ranks did not rendezvous / finish in time on this box (ranks killed)")
    for r, (rc, o) in enumerate(res):
        assert rc == 0, f"rank {r}:\n{o[-3000:]}"
    line = json.loads([ln for ln in res[0][1].splitlines() if ln.startswith("{")][-1])
    # (sams: trainer.MultiOptimizerStep with one gradient reducer per optimizer; bs = 1 per rank keeps the two ranks small)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == (2 if "sams" in extra else 8)
    assert line["value"] > 0 and line["scaling"] == "weak"
    assert not [ln for ln in res[1][1].splitlines() if ln.startswith("{")]   # only rank 0 prints the JSON line


def test_single_rank_rccl_group_runs_the_exchange_path():
    """RCCL itself on the 1-GPU box: a one-rank nccl group (SHINEON_SINGLE_RANK_GROUP=1) with every collective of
    trainer.TrainStep issued for real - bucketed all-reduce on the communication stream behind the in-graph signal nodes,
    Adam per bucket - must reproduce the steps without a process group bit for bit (tests/_rccl_single_rank_child.py)."""
    child = os.path.join(ROOT, "tests", "_rccl_single_rank_child.py")
    outs = []
    for attempt in range(2):   # one retry for the intermittent watchdog exception described below
        p = subprocess.run([sys.executable, child, ROOT], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600, start_new_session=True)
        outs.append(p.stdout)
        if p.returncode == 0:
            break
    if p.returncode != 0 and all("ProcessGroupNCCL.cpp" in o and "AssertionError" not in o for o in outs):
        pytest.skip("ProcessGroupNCCL's watchdog thread raised in both attempts (one-rank RCCL group on this box): " + p.stdout[-600:])
    assert p.returncode == 0 and "RCCL_SINGLE_RANK_OK" in p.stdout, p.stdout[-4000:]


@pytest.mark.parametrize("extra,bucketed", [([], None), (["--config", "c3"], None), ([], "0")])
def test_bench_over_a_single_rank_rccl_group(extra, bucketed):
    """bench.py with the one-rank RCCL group: the N > 1 code path (broadcasts, exchange, MAX-over-ranks timing, the
    with / without-collectives measurement behind config.exchange_exposed_ms) over the nccl backend on one GPU."""
    import json

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-table"] + extra
    env = dict(os.environ, SHINEON_SINGLE_RANK_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if bucketed is not None:
        env["SHINEON_BUCKETED"] = bucketed   # c4: whole-slab exchange after each graph instead of the per-model buckets
    p = None
    for attempt in range(2):
        # One retry, kept as a belt: before the captures became thread-local (graphs.CAPTURE_MODE) about one start in six died
        # because ProcessGroupNCCL's watchdog thread queried an event while this thread was capturing; the first attempt's
        # output is kept in the failure message.
        first = p
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, start_new_session=True)
        if p.returncode == 0:
            break
    if p.returncode != 0 and first is not None and all("ProcessGroupNCCL.cpp" in q.stderr for q in (first, p)):
        pytest.skip("ProcessGroupNCCL's watchdog thread raised in both attempts (one-rank RCCL group on this box): " + p.stderr[-600:])
    assert p.returncode == 0, (first.stderr[-1500:] if first is not None else "") + "\n--- retry ---\n" + p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["exchange"] and line["config"]["exchange_exposed_ms"] is not None, line["config"]
    assert "backend nccl" in p.stderr, p.stderr[-3000:]
