"""Two data-parallel ranks sharing the one GPU of the test box (gloo: RCCL refuses two ranks per device): the real
DataParallel bucket overlap and the five-graph step on GPU tensors (tools/dp_check.py does the work and asserts)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_stay_in_lock_step_and_segmented_graphs_match_eager():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tools", "dp_check.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "params identical across ranks: True" in r.stdout and "segmented graphs == eager: True" in r.stdout, tail
