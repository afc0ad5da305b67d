"""Data-parallel wrapper on CPU: 2 processes, gloo backend (the GPU build uses RCCL through the same API)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from mmfn_amd.parallel import DataParallel
from mmfn_amd.params import FlatLayout


class _Tiny(nn.Module):
    """Parameter names that hit every backward stage + a never-trained tail tensor."""

    def __init__(self):
        super().__init__()
        self.encoder = nn.Module()
        self.encoder.transformer4 = nn.Linear(8, 8)
        self.encoder.layer3 = nn.Conv2d(4, 4, 3, bias=False)
        self.encoder.transformer2 = nn.Linear(6, 3)
        self.encoder.stem = nn.Linear(5, 7)
        self.encoder.unused = nn.Linear(3, 3)
        self.join = nn.Linear(4, 2)
        self.bn = nn.BatchNorm2d(4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
    m = _Tiny()
    object.__setattr__(m, "_layout", FlatLayout(m, ("encoder.unused.weight", "encoder.unused.bias")).materialize("cpu"))
    L = m._layout
    dp = DataParallel(m, dist, max_bucket_bytes=64)  # tiny buckets: several chunks per stage
    dp.broadcast_parameters()
    ref = [torch.empty_like(L.params) for _ in range(world)]
    dist.all_gather(ref, L.params)
    same_params = all(torch.equal(ref[0], r) for r in ref)
    # buckets tile [0, tail) exactly
    spans = sorted(c for chunks in dp.buckets for c in chunks)
    covered = spans[0][0] == 0 and spans[-1][1] == L.tail and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    L.grads.copy_(torch.arange(L.total, dtype=torch.float32) * (rank + 1))
    tail_before = L.grads[L.tail:].clone()
    for stage in range(4):
        dp.on_stage(stage)
    dp.finish()
    expect = torch.arange(L.total, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok_sum = torch.equal(L.grads[:L.tail], expect[:L.tail])
    ok_tail = torch.equal(L.grads[L.tail:], tail_before)
    # p.grad views see the reduced values (reference-style optimizers read p.grad)
    L.attach_grads()
    g = m.encoder.layer3.weight.grad
    off, n = L.offsets["encoder.layer3.weight"]
    ok_view = torch.equal(g.permute(0, 2, 3, 1).reshape(-1), expect[off:off + n]) and m.encoder.unused.weight.grad is None
    # bf16 buckets (the bf16 training mode's wire format): cast -> sum -> cast back into the fp32 buffer; small integers are exact
    dp16 = DataParallel(m, dist, max_bucket_bytes=64, grad_dtype="bf16")
    vals = (torch.arange(L.total) % 61).float()
    L.grads.copy_(vals * (rank + 1) + 0.001953125 * rank)     # rank 1 carries a fraction that bf16 drops: 2^-9
    tail_before = L.grads[L.tail:].clone()
    dp16.begin()
    for stage in range(4):
        dp16.on_stage(stage)
    dp16.finish()
    want = sum((vals * (r + 1) + 0.001953125 * r).bfloat16().float() for r in range(world)).bfloat16().float()
    ok_sum16 = torch.equal(L.grads[:L.tail], want[:L.tail]) and torch.equal(L.grads[L.tail:], tail_before) \
        and dp16.bytes_per_step() * 2 == dp.bytes_per_step() and L.grads.dtype == torch.float32
    if rank == 0:
        out.put((same_params, covered, ok_sum, ok_tail, ok_view, dp.world, ok_sum16))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    same_params, covered, ok_sum, ok_tail, ok_view, world, ok_sum16 = res
    assert world == 2
    assert same_params, "rank-0 broadcast did not equalise the parameters"
    assert covered, "gradient buckets must tile the trained range exactly"
    assert ok_sum, "all-reduce (sum) over the trained range"
    assert ok_tail, "never-trained tail must be excluded from the reduction"
    assert ok_view, "p.grad views alias the reduced flat buffer"
    assert ok_sum16, "bf16 buckets: the fp32 gradient buffer must hold bf16(sum of the bf16-rounded buckets); half the bytes"


def test_stage_order_of_real_model():
    """Flat storage is ordered by backward stage: deepest fusion scale first, stem/VectorNet last."""
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    m = MMFN(GlobalConfig(), "cpu")
    L = m._layout
    r = L.stage_ranges
    assert r[0][0] == 0 and all(r[i][1] == r[i + 1][0] for i in range(3)) and r[3][1] == L.tail
    assert L.offsets["encoder.transformer4.blocks.0.mlp.0.weight"][0] < r[0][1]
    assert r[1][0] <= L.offsets["encoder.image_encoder.features.layer3.0.conv1.weight"][0] < r[1][1]
    assert r[3][0] <= L.offsets["encoder.vectornet_encoder.generator.3.weight"][0] < r[3][1]
    sizes = [(e - b) * 4 / 2 ** 20 for b, e in r]
    assert sizes[0] > sum(sizes[1:])  # the first bucket (scale 4) is the largest -> most overlap


def _resume_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmfn_amd.trainer import Trainer, sync_resume_state

    class _Opt(object):
        def __init__(self):
            self.param_groups = [dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, params=[]),
                                 dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, params=[])]

    tr, opt = Trainer("cpu", None), _Opt()
    if rank == 0:  # what Trainer.resume() restored on rank 0 only
        tr.cur_epoch, tr.cur_iter, tr.bestval, tr.bestval_epoch = 7, 1234, 0.25, 5
        tr.train_loss, tr.val_loss = [3.0, 2.0], [2.5]
        opt.param_groups[0].update(lr=2.5e-5, betas=(0.8, 0.99))
        opt.param_groups[1].update(lr=5e-5)
    sync_resume_state(tr, opt, dist)
    got = (tr.cur_epoch, tr.cur_iter, tr.bestval, tr.bestval_epoch, tr.train_loss, tr.val_loss,
           opt.param_groups[0]["lr"], opt.param_groups[0]["betas"], opt.param_groups[1]["lr"], opt.param_groups[1]["weight_decay"])
    # every rank now iterates the same epoch range
    out.put((rank, got, list(range(tr.cur_epoch, 9))))
    dist.barrier()
    dist.destroy_process_group()


def test_resume_state_reaches_every_rank_gloo():
    """after a rank-0 resume the other ranks must take its epoch counter and learning rate, otherwise
    they iterate different epoch ranges and deadlock in the gradient all-reduce."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_resume_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2] == [7, 8]
    assert res[1][1][:4] == (7, 1234, 0.25, 5) and res[1][1][6] == 2.5e-5 and res[1][1][7] == (0.8, 0.99)
