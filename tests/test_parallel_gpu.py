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
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", MMFN_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0",
               DP_CHECK_BATCH="32")   # the benched per-GPU batch
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "tools", "dp_check.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=3000)
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
    # capturable: the collective is plain stream work.  (The capture is owned by mmfn_amd.graphs.Graph like every capture of the
    # package: destroyed at a safe point - parked, then RcclComm.destroy() drains the parked ones before the communicator goes.
    # A bare torch.cuda.CUDAGraph destroyed right here made a LATER, unrelated replay die inside hipGraphLaunch on ROCm 7.0:
    # tools/experiments/segv_bisect2.sh.)
    from mmfn_amd import graphs
    y = torch.ones(4096, device="cuda:0")
    torch.cuda.synchronize()
    g = graphs.Graph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        g.capture_begin(capture_error_mode="thread_local")
        y.mul_(2.0)
        c.all_reduce_sum_(y)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(cap)
    g.replay(); g.replay()
    torch.cuda.synchronize()
    assert float(y[0]) == 4.0           # (capture does not execute; two replays double twice)
    del g
    c.destroy()


class _OneRank(object):
    """torch.distributed stand-in for a 1-rank world (the C-ABI transport carries the collectives)."""

    class ReduceOp(object):
        SUM = 0

    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def get_rank():
        return 0

    @staticmethod
    def broadcast(t, src):
        return None


def test_single_graph_data_parallel_step_with_captured_collectives():
    """The step the C-ABI transport enables: forward + backward + 17 gradient-bucket all-reduces (issued from the branch lanes'
    own streams onto the communication stream) + AdamW captured into ONE hipGraph.  One GPU -> a 1-rank communicator (the sum
    over one rank is the identity), so the replayed step must equal the plain single-GPU graph bit for bit - which it only
    does if every bucket's dependency (wait for its lane, rejoin before AdamW) was captured correctly."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from mmfn_amd.comm import RcclComm
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    from mmfn_amd.parallel import DataParallel, GraphedStep
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    inp, gt = bench.synth_inputs(2, dev, seed=3, lanes=16, n_lidar=4096)
    outs = []
    for use_dp in (False, True):
        torch.manual_seed(5)
        net = MMFN(GlobalConfig(), dev)
        net.train()
        eng = net._engine_for()
        dp = None
        if use_dp:
            comm = RcclComm(0, 1)
            dp = DataParallel(net, _OneRank, comm=comm, max_bucket_bytes=16 << 20)
            assert dp.n_buckets() >= 17
        eng.train_step(inp, gt, dp=dp)
        step = GraphedStep(eng, dp, inp, gt, warm=0)
        if use_dp:
            assert step.single_graph and step.recorder.n_graphs == 1
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        outs.append((net._layout.params.clone(), float(step.loss.item())))
        if use_dp:
            with pytest.raises(Exception, match="still hold collectives"):   # a communicator must outlive the captures that use it
                comm.destroy()
            del step
            comm.destroy()
    assert outs[0][1] == outs[1][1]
    assert torch.equal(outs[0][0], outs[1][0])


def test_bf16_gradient_buckets_through_the_c_abi_and_inside_one_graph():
    """mmfn_allreduce_sum_bf16 (include/mmfn_comm.h): the bf16 training mode's buckets are cast to bf16, summed by RCCL, and
    cast back into the fp32 gradient buffer on the communication stream.  1-rank communicator: the 'sum' is the identity, so
    after the exchange the gradient buffer must hold exactly the bf16 rounding of the local gradients (and the never-trained
    tail untouched); the whole step - forward, backward, 17+ (cast, all-reduce, cast) triples, AdamW - captures into ONE hipGraph
    whose replays equal eager data-parallel steps bit for bit.  (One network, one DataParallel, one communication stream
    throughout: the communicator is only ever driven from that stream.)"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from mmfn_amd import ops
    from mmfn_amd.comm import RcclComm
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    from mmfn_amd.parallel import DataParallel, GraphedStep
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    inp, gt = bench.synth_inputs(2, dev, seed=3, lanes=16, n_lidar=4096)
    comm = RcclComm(0, 1)
    torch.manual_seed(5)
    net = MMFN(GlobalConfig(act_dtype="bf16"), dev)
    net.train()
    assert DataParallel(net, _OneRank, max_bucket_bytes=16 << 20).grad_dtype == "f32"   # the default in every mode: what the reference's DDP exchanges
    dp = DataParallel(net, _OneRank, comm=comm, max_bucket_bytes=16 << 20, grad_dtype="bf16")   # the opt-in
    eng, L = net._engine_for(), net._layout
    assert dp.grad_dtype == "bf16" and dp.bytes_per_step() == 2 * L.tail and dp.n_buckets() >= 17
    init = {k: v.detach().clone() for k, v in net.state_dict().items()}
    rng0, step0 = eng.rng_state.clone(), eng.step_count.clone()

    def reset():
        net.load_state_dict(init)
        eng.rng_state.copy_(rng0)
        eng.step_count.copy_(step0)
        L.exp_avg.zero_()
        L.exp_avg_sq.zero_()

    # ---- the exchange itself
    ops.rng_advance(eng.rng_state)
    eng.forward(inp, True, gt)
    eng.backward()
    torch.cuda.synchronize()
    local = L.grads.clone()
    dp.begin()
    for key in dp.group_order:
        dp.reduce(key)
    dp.finish()
    torch.cuda.synchronize()
    assert torch.equal(L.grads[:L.tail], local[:L.tail].bfloat16().float()) and torch.equal(L.grads[L.tail:], local[L.tail:])
    assert not torch.equal(L.grads[:L.tail], local[:L.tail])
    # ---- three eager data-parallel steps ...
    reset()
    for _ in range(3):
        la = eng.train_step(inp, gt, dp=dp)
    torch.cuda.synchronize()
    want, la = L.params.clone(), float(la.item())
    # ... == one eager step + two replays of the single captured graph
    reset()
    step = GraphedStep(eng, dp, inp, gt, warm=1)
    assert step.single_graph and step.recorder.n_graphs == 1
    for _ in range(2):
        lb = step()
    torch.cuda.synchronize()
    assert la == float(lb.item())
    assert torch.equal(want, L.params)
    del step
    comm.destroy()


@pytest.mark.parametrize("dtype", ["bf16"])   # (f32 on the same data-parallel path: the lock-step test above; each run moves 419 MB
def test_bench_launches_its_own_ranks(dtype):  # of gradients per step through gloo on the host, ~4 minutes on a slow box)
    """`python bench.py --gpus 2` with no outer launcher (how the driver starts the single-GPU bench): bench.py starts the two
    ranks itself and rank 0 prints the JSON line with the `comm` block.  One GPU here, so both ranks share cuda:0 over gloo
    (MMFN_BENCH_SINGLE_DEVICE); on a multi-GPU node the same command runs one rank per GPU over RCCL.  dtype bf16 = BASELINE
    configs[2]'s arithmetic (bf16 training mode, 32 samples per rank) on the data-parallel path."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MMFN_BENCH_SINGLE_DEVICE="1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline",
           "--no-oracle-check", "--profile-steps", "1", "--dtype", dtype, "--grad-dtype", "bf16"]   # the bf16-on-the-wire opt-in
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, tail
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 64 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert rec["dtype"] == dtype
    c = rec["comm"]
    assert c["ranks"] == 2 and c["buckets"] >= 10 and c["ranks_in_lock_step"] is True and c["exposed_ms_per_step"] >= 0.0
    # --grad-dtype bf16: the gradient buckets cross the wire as bf16, 2 bytes per trained parameter (the default, fp32 as under the
    # reference's DDP: 4 = 419 MB, asserted in test_bf16_gradient_buckets_through_the_c_abi_and_inside_one_graph and on CPU)
    assert c["gradient_dtype_on_the_wire"] == "bf16"
    assert 200e6 < c["allreduce_bytes_per_step"] < 300e6 and rec["config"]["hipgraph"] is True
