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


CHILD_TIMEOUT_S = 180   # every child of this file: a hang is a FAILURE within three minutes (never a skip, never a retry)


def _spawn_pair(cmd, env, timeout=CHILD_TIMEOUT_S):
    """Start the two ranks as fresh child processes and wait.  Returns [(returncode, output)]; a pair that does not finish in
    time is killed (nothing is left on the GPU) and the test FAILS with what the ranks printed so far."""
    env = dict(env, GLOO_SOCKET_IFNAME=env.get("GLOO_SOCKET_IFNAME", "lo"), SHINEON_DIST_TIMEOUT_S="120")
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, start_new_session=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        tails = [p.communicate()[0] for p in procs]
        pytest.fail(f"the two ranks did not finish within {timeout} s (killed):\n" + "\n--- other rank ---\n".join((t or "")[-2000:] for t in tails))
    return [(p.returncode, o) for p, o in zip(procs, outs)]


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run_children(env):
    child = os.path.join(ROOT, "tests", "_dp_gpu_child.py")
    res = _spawn_pair([sys.executable, child, ROOT], dict(env, MASTER_PORT=_free_port()))
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
    for r, (rc, o) in enumerate(res):
        assert rc == 0, f"rank {r}:\n{o[-3000:]}"
    line = json.loads([ln for ln in res[0][1].splitlines() if ln.startswith("{")][-1])
    # (sams: trainer.MultiOptimizerStep with one gradient reducer per optimizer; bs = 1 per rank keeps the two ranks small)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == (2 if "sams" in extra else 8)
    assert line["value"] > 0 and line["scaling"] == "weak"
    assert not [ln for ln in res[1][1].splitlines() if ln.startswith("{")]   # only rank 0 prints the JSON line


def test_bench_gpus_2_without_a_launcher_starts_two_ranks(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start two ranks itself (fresh child processes
    through torch.distributed.run, decided before the parent touches the GPU) and relay rank 0's line - not silently
    benchmark one GPU.  Both ranks on cuda:0 over gloo here (SHINEON_LOCAL_DEVICE also tells bench.py that ranks may share a
    device)."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SHINEON_DIST_BACKEND="gloo", SHINEON_LOCAL_DEVICE="0", GLOO_SOCKET_IFNAME="lo", SHINEON_DIST_TIMEOUT_S="120")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-hbm-table", "--config", "c2"]
    p = _run_one(cmd, env)
    assert p.returncode == 0, p.stderr[-4000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["config"]["global_batch"] == 8 and line["value"] > 0
    assert len([ln for ln in p.stdout.splitlines() if ln.startswith("{")]) == 1   # only rank 0 prints


def test_single_rank_rccl_group_runs_the_exchange_path():
    """RCCL itself on the 1-GPU box: a one-rank nccl group (SHINEON_SINGLE_RANK_GROUP=1) with every collective of
    trainer.TrainStep issued for real - bucketed all-reduce on the communication stream behind the in-graph signal nodes,
    Adam per bucket - must reproduce the steps without a process group bit for bit (tests/_rccl_single_rank_child.py)."""
    child = os.path.join(ROOT, "tests", "_rccl_single_rank_child.py")
    p = _run_one([sys.executable, child, ROOT], dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SHINEON_DIST_TIMEOUT_S="120"))
    assert p.returncode == 0 and "RCCL_SINGLE_RANK_OK" in p.stdout, p.stdout[-4000:] + "\n" + p.stderr[-2000:]


def _run_one(cmd, env):
    """One child, one attempt: any crash - ProcessGroupNCCL's watchdog included - and any hang (killed after CHILD_TIMEOUT_S,
    below the process group's own time-out so that the kill is not pre-empted by a watchdog message) fails the test."""
    try:
        return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=CHILD_TIMEOUT_S,
                              start_new_session=True)
    except subprocess.TimeoutExpired as e:
        out = (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or "")
        err = (e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
        pytest.fail(f"{' '.join(cmd[-6:])} did not finish within {CHILD_TIMEOUT_S} s (killed):\n{out[-1500:]}\n{err[-2500:]}")


@pytest.mark.parametrize("extra,bucketed", [([], None), (["--config", "c3"], None), ([], "1")])
def test_bench_over_a_single_rank_rccl_group(extra, bucketed):
    """bench.py with the one-rank RCCL group: the N > 1 code path (broadcasts, exchange, MAX-over-ranks timing, the
    with / without-collectives measurement behind config.exchange_exposed_ms) over the nccl backend on one GPU."""
    import json

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-table"] + extra
    env = dict(os.environ, SHINEON_SINGLE_RANK_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if bucketed is not None:
        env["SHINEON_BUCKETED"] = bucketed   # c4: per-bucket exchange released inside each model's backward (default: whole slab)
    env["SHINEON_DIST_TIMEOUT_S"] = "120"
    p = _run_one(cmd, env)
    assert p.returncode == 0, p.stderr[-4000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["exchange"] and line["config"]["exchange_exposed_ms"] is not None, line["config"]
    assert "backend nccl" in p.stderr, p.stderr[-3000:]


def test_train_py_and_test_py_entry_points_end_to_end(tmp_path):
    """`python train.py ...` then `python test.py --checkpoint ...` as the reference's users run them (train.py:32-141,
    test.py:9-10), on the synthetic dataset: training writes checkpoints/final.ckpt under experiments_dir/name, the test run
    loads it through load_from_checkpoint + override_hparams and writes the warp-cloth PNGs the next stage reads
    under result_dir/name/<ckpt>/<datamode>/<Dataset>/.  A run that dies leaves checkpoints/interrupted_by_<Exception>.ckpt
    and exits non-zero (the reference's save_on_interrupt, train.py:121-137)."""
    root, res = str(tmp_path / "exp"), str(tmp_path / "res")
    common = ["--model", "warp", "--dataset", "synthetic", "--name", "cli", "--batch", "2", "--workers", "0", "--synthetic_length", "8"]
    env = dict(os.environ)
    p = _run_one([sys.executable, os.path.join(ROOT, "train.py")] + common +
                 ["--experiments_dir", root, "--keep_epochs", "1", "--decay_epochs", "0", "--limit_train_batches", "3",
                  "--limit_val_batches", "1", "--val_check_interval", "2"], env)
    assert p.returncode == 0, p.stderr[-3000:]
    ckpt = os.path.join(root, "cli", "checkpoints", "final.ckpt")
    assert os.path.exists(ckpt), os.listdir(os.path.join(root, "cli"))
    p = _run_one([sys.executable, os.path.join(ROOT, "test.py")] + common + ["--checkpoint", ckpt, "--result_dir", res], env)
    assert p.returncode == 0 and "Testing........" in p.stdout, p.stderr[-3000:]
    out = os.path.join(res, "cli", "final.ckpt", "test", "SyntheticDataset")
    # (warp-mask is only written for VitonDataset: the reference's rule, util/visualization.py:60-88, mirrored in io_png.save_images)
    assert sorted(os.listdir(out)) == ["warp-cloth"] and len(os.listdir(os.path.join(out, "warp-cloth"))) == 8
    # a failing run: the exception is logged, the checkpoint carries its name, the exit code is not 0
    bad = _run_one([sys.executable, "-c",
                    "import sys; sys.path.insert(0, %r); import shineon_virtual_tryon_amd\n"
                    "from shineon_virtual_tryon_amd import cli, warp_model\n"
                    "calls = [0]\n"
                    "orig = warp_model.WarpModel.training_step\n"
                    "def boom(self, *a, **k):\n"
                    "    calls[0] += 1\n"
                    "    if calls[0] > 8: raise ValueError('injected')\n"
                    "    return orig(self, *a, **k)\n"
                    "warp_model.WarpModel.training_step = boom\n"
                    "sys.exit(cli.main(True, %r))" % (ROOT, common + ["--experiments_dir", root + "2", "--no_shuffle"])], env)
    assert bad.returncode == 1, (bad.returncode, bad.stderr[-2000:])
    assert os.path.exists(os.path.join(root + "2", "cli", "checkpoints", "interrupted_by_ValueError.ckpt")), bad.stderr[-2000:]
