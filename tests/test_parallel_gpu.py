"""Two data-parallel ranks sharing the one GPU of the test box (gloo: RCCL refuses two ranks per device): the real
DataParallel bucket overlap and the graph-replayed step on GPU tensors (tools/dp_check.py does the work and asserts)."""
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


def test_rccl_c_abi_single_rank_communicator():
    """The C-ABI transport (libmmfn_comm.so -> RCCL) on the one GPU of the test box: a 1-rank communicator created through
    mmfn_comm_*, gradient buckets reduced through mmfn_allreduce_sum_f32 on a side stream via DataParallel(comm=...), and the
    same calls captured into a hipGraph.  (More than one rank needs more than one GPU: tools/dp_check.py, MMFN_DP_TRANSPORT.)"""
    import torch
    from mmfn_amd.comm import RcclComm
    torch.cuda.set_device(0)
    c = RcclComm(0, 1)
    assert c.ranks() == (1, 0)
    x = torch.arange(1 << 20, dtype=torch.float32, device="cuda:0")
    ref = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    c.all_reduce_sum_(x, stream=side)
    c.broadcast_(x, root=0, stream=side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)          # sum over one rank
    # capturable: the collective is plain stream work
    g = torch.cuda.CUDAGraph()
    y = torch.ones(4096, device="cuda:0")
    torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y.mul_(2.0)
        c.all_reduce_sum_(y)
    g.replay(); g.replay()
    torch.cuda.synchronize()
    assert float(y[0]) in (4.0, 8.0)    # capture itself may or may not execute the work once; replays double twice
    c.destroy()
