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


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_ranks_equal_one_rank_and_stay_in_sync():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    child = os.path.join(ROOT, "tests", "_dp_gpu_child.py")
    procs = [subprocess.Popen([sys.executable, child, ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"DP_ALL_OK {r}" in o, o[-4000:]
