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
    # the VALUE of the reduced gradient: mean over ranks of the per-rank CPU-oracle gradients; weights after the step equal a
    # reference-style 2-rank loop (torch AdamW on the averaged gradient)
    assert "reduced gradient == mean of per-rank oracle gradients: True" in r.stdout, tail


def test_two_ranks_over_rccl():
    """The production transport: one rank per GPU over RCCL.  Needs two devices; the round's test box has one, so this is
    skipped LOUDLY there - tools/dp_check.py picks nccl by itself whenever enough devices are visible."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL PATH NOT EXERCISED: %d GPU visible, two ranks need two devices (RCCL refuses two ranks per device)"
                    % torch.cuda.device_count())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", MMFN_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "tools", "dp_check.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "backend: nccl" in r.stdout and "reduced gradient == mean of per-rank oracle gradients: True" in r.stdout, tail
