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
    if world == 2:   # one addition: exactly bf16(a + b)
        ok16 = torch.equal(L.grads[:L.tail], want[:L.tail])
    else:            # more ranks: the ring rounds to bf16 at every hop, in an order that is the backend's - bounded, not bit-defined
        mag = sum((vals * (r + 1) + 0.001953125 * r).abs() for r in range(world))
        exact = sum((vals * (r + 1) + 0.001953125 * r).double() for r in range(world))
        ok16 = bool(((L.grads[:L.tail].double() - exact[:L.tail]).abs() <= world * 2.0 ** -8 * mag[:L.tail].double() + 1e-30).all())   # one rounding per addend + one per hop
    ok_sum16 = ok16 and torch.equal(L.grads[L.tail:], tail_before) \
        and dp16.bytes_per_step() * 2 == dp.bytes_per_step() and L.grads.dtype == torch.float32
    if rank == 0:
        out.put((same_params, covered, ok_sum, ok_tail, ok_view, dp.world, ok_sum16))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 8])   # 8 = the ranks of BASELINE configs[2] (one node of 8 GPUs)
def test_bucketed_allreduce_two_ranks_gloo(world_size):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world_size, port, out)) for r in range(world_size)]
    for p in procs:
        p.start()
    res = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    same_params, covered, ok_sum, ok_tail, ok_view, world, ok_sum16 = res
    assert world == world_size
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


def _bf16_error_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _Tiny()
    object.__setattr__(m, "_layout", FlatLayout(m, ("encoder.unused.weight", "encoder.unused.bias")).materialize("cpu"))
    L = m._layout
    default_is_f32 = DataParallel(m, dist).grad_dtype == "f32"
    g = torch.Generator().manual_seed(7 + rank)
    # gradients of very different magnitude per element (per-parameter gradient norms span 0 .. 1e5)
    mine = torch.randn(L.total, generator=g) * torch.logspace(-6, 4, L.total)
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    res = {}
    for kind in ("f32", "bf16"):
        dp = DataParallel(m, dist, max_bucket_bytes=64, grad_dtype=kind)
        L.grads.copy_(mine)
        dp.begin()
        for stage in range(4):
            dp.on_stage(stage)
        dp.finish()
        res[kind] = L.grads[:L.tail].clone()
    exact = sum(b.double() for b in both)[:L.tail]
    mag = sum(b.abs().double() for b in both)[:L.tail]
    err32 = (res["f32"].double() - exact).abs()
    err16 = (res["bf16"].double() - exact).abs()
    # bf16 on the wire: every addend rounded once (2^-8 relative: 8 significant bits), the sum rounded once more -> |error| <= 2^-7 * sum |addend|
    bounded = bool((err16 <= 2.0 ** -7 * mag + 1e-30).all())
    exact32 = bool((err32 <= 2.0 ** -24 * mag + 1e-30).all())
    cos = float(torch.nn.functional.cosine_similarity(res["bf16"].double(), exact, dim=0))
    if rank == 0:
        out.put((default_is_f32, bounded, exact32, cos, float((err16 / (mag + 1e-30)).max())))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_bucket_error_is_bounded_against_the_fp32_exchange():
    """bf16 on the wire is an opt-in, not the bf16 mode's default (the reference's DDP under autocast all-reduces
    fp32 gradients, phase2_train_net.py:265-269), and its error against the fp32 exchange is bounded: 2^-7 of the addends'
    magnitudes per element, direction of the whole buffer kept to 1e-5."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_error_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    default_is_f32, bounded, exact32, cos, worst = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert default_is_f32, "DataParallel must exchange fp32 gradients unless bf16 is asked for"
    assert exact32, "the fp32 exchange is the exact sum to fp32 rounding"
    assert bounded, "bf16 buckets: |error| must stay below 2^-7 * sum |addend| per element (worst seen %.3g)" % worst
    assert cos > 1.0 - 1e-5, cos


class _FakeComm(object):
    """Stands in for comm.RcclComm in the CPU tests of the transport-selection protocol."""
    destroyed = None   # a multiprocessing list shared with the test: (rank, "destroyed") records

    def __init__(self, rank):
        self.rank = rank

    def destroy(self):
        _FakeComm.destroyed.append(self.rank)


def _fallback_worker(rank, world, port, out, destroyed, scenario):
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmfn_amd import comm as C
    from mmfn_amd.parallel import connect
    _FakeComm.destroyed = destroyed
    m = _Tiny()
    object.__setattr__(m, "_layout", FlatLayout(m, ("encoder.unused.weight", "encoder.unused.bias")).materialize("cpu"))

    def make_id(r):
        if scenario == "no-library-on-rank-1" and r == 1:
            raise C.MMFNCommError("libmmfn_comm.so not found")
        return (b"\x01" if r == 0 else b"\x00") * C.ID_BYTES

    seen_id = []

    def make_comm(r, w, unique_id, dev):
        seen_id.append(unique_id)
        if scenario == "init-times-out-on-rank-1" and r == 1:
            time.sleep(2.5)   # mmfn_comm_init sitting in RCCL's bootstrap long past the limit ...
            return _FakeComm(r)   # ... and coming back after this rank has voted no
        if scenario == "init-errors-on-rank-0" and r == 0:
            raise C.MMFNCommError("mmfn_comm_init failed with code 2")
        return _FakeComm(r)

    t0 = time.time()
    dp, note = connect(m, dist, transport="auto", transport_opts=dict(timeout_s=0.5, make_id=make_id, make_comm=make_comm))
    took = time.time() - t0
    # the launcher's group must still be usable, in step on every rank: the next collective returns the right sum
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    time.sleep(3.0 if scenario == "init-times-out-on-rank-1" else 0.1)   # let the abandoned helper finish
    dist.barrier()
    if rank == 0:
        out.put((dp.comm is None, note, float(t.item()), took, sorted(destroyed), [i[:1] for i in seen_id]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["init-times-out-on-rank-1", "init-errors-on-rank-0", "no-library-on-rank-1", "all-fine"])
def test_connect_falls_back_when_the_c_abi_transport_fails_on_one_rank_only(scenario):
    """The asymmetric failures that would hang a first real 8-rank run (ADVICE r4 comm.py:128): the
    communicator init never returns on ONE rank / errors on one rank / the library is missing on one rank.  Every rank must end
    up on torch.distributed, the launcher's process group must still be in step (a following all-reduce is right), no communicator
    may be left alive (the one that came up is destroyed, the late one destroys itself), and nobody waits longer than the limit."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    mgr = ctx.Manager()
    destroyed = mgr.list()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, out, destroyed, scenario)) for r in range(2)]
    for p in procs:
        p.start()
    fell_back, note, total, took, gone, ids = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert total == 3.0, "the process group is out of step after the transport selection"
    if scenario == "all-fine":
        assert not fell_back and note is None and gone == []
        assert ids == [b"\x01"], "every rank must receive rank 0's rendezvous id"
        return
    assert fell_back and note.startswith("fallback:"), (fell_back, note)
    assert took < 2.0, "rank 0 waited %.1f s for a decision with a 0.5 s limit" % took
    if scenario == "init-times-out-on-rank-1":
        assert gone == [0, 1], "rank 0's communicator and rank 1's late one must both be destroyed, got %r" % (gone,)
    elif scenario == "init-errors-on-rank-0":
        assert gone == [1], gone
    else:
        assert gone == [] and ids == [], "without the library on one rank nobody may get as far as the id broadcast / init"


def _order_worker(rank, world, port, group_ranges, tail, program, out):
    """One rank of test_bucket_issue_order_is_the_program_order_on_every_rank: DataParallel over a stand-in layout with the REAL
    model's 17 readiness groups (ranges scaled down), reports arriving with rank-dependent delays."""
    import random
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class _Layout(object):
        pass

    class _Mod(object):
        pass

    L = _Layout()
    L.group_ranges, L.tail, L.device = group_ranges, tail, torch.device("cpu")
    L.grads = torch.arange(tail + 8, dtype=torch.float32) * (rank + 1)
    m = _Mod()
    m._layout = L
    dp = DataParallel(m, dist, max_bucket_bytes=1 << 10)
    issued = []
    real = dist.all_reduce

    class _Logged(object):   # torch.distributed with all_reduce logging the range it was called on
        ReduceOp = dist.ReduceOp

        @staticmethod
        def get_world_size():
            return world

        @staticmethod
        def all_reduce(t, op=None, async_op=False):
            issued.append((t.storage_offset(), t.numel()))
            return real(t, op=op, async_op=async_op)

    dp.dist = _Logged
    rnd = random.Random(1000 + rank)
    dp.begin()
    for key in program:
        time.sleep(rnd.random() * 0.02)    # this rank's lanes finish (are enqueued) at their own pace
        dp.reduce(key)
    dp.finish()
    want = torch.arange(tail + 8, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(L.grads[:tail], want[:tail])
    logs = [None] * world
    dist.all_gather_object(logs, issued)
    if rank == 0:
        out.put((ok, logs, dp.n_buckets(), len(dp.group_order)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_issue_order_is_the_program_order_on_every_rank():
    """The one hazard a real 8-rank RCCL run adds over gloo: collectives on one communicator must be issued in the same order on
    every rank, whatever order the GPUs finish their lanes in.  DataParallel issues a bucket where the ENGINE'S HOST PROGRAM
    reports it (Engine.backward_scale: transformer, then the lanes in the order Engine._branches enqueues them), never from a
    completion callback - so with 8 ranks whose reports arrive with different delays the logged all-reduce sequences are identical,
    chunk for chunk, and equal to the program's; a rank that reported two lanes the other way round would be caught by this check."""
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    L = MMFN(GlobalConfig(), "cpu")._layout
    # the real 17 groups, each range scaled to a few hundred elements (order, adjacency and 4-alignment kept)
    keys = sorted(L.group_ranges, key=lambda k: L.group_ranges[k][0])
    assert len(keys) == 17
    ranges, pos = {}, 0
    for k in keys:
        b, e = L.group_ranges[k]
        n = max(8, ((e - b) // 65536 + 3) // 4 * 4)
        ranges[k] = (pos, pos + n)
        pos += n
    # the engine's host order: per backward stage the transformer / head groups first, then the lanes as _branches enqueues them
    # (side lanes before the main lane)
    program = []
    for st in range(4):
        ks = [k for k in keys if k[0] == st]
        lanes = [k for k in ks if k[1] in ("img", "lid", "map", "vec")]
        program += [k for k in ks if k not in lanes] + [k for k in lanes if k[1] != "img"] + [k for k in lanes if k[1] == "img"]
    assert sorted(program) == sorted(keys)
    world = 8
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_order_worker, args=(r, world, port, ranges, pos, program, out)) for r in range(world)]
    for p in procs:
        p.start()
    ok, logs, n_buckets, n_groups = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok, "sums over the 17 groups"
    assert n_groups == 17 and len(logs[0]) == n_buckets >= 17
    assert all(l == logs[0] for l in logs), "every rank must issue the same all-reduce sequence"
    expect = [(b0, min(b0 + 256, e) - b0) for k in program for b, e in [ranges[k]] for b0 in range(b, e, 256)]
    assert logs[0] == expect, "the sequence is the program's: group by group, chunk by chunk"
    swapped = list(program)
    i = next(i for i, k in enumerate(swapped) if k[1] == "lid")
    swapped[i], swapped[i + 1] = swapped[i + 1], swapped[i]
    other = [(b0, min(b0 + 256, e) - b0) for k in swapped for b, e in [ranges[k]] for b0 in range(b, e, 256)]
    assert other != expect, "a rank reporting two lanes the other way round issues a different sequence (this check can fail)"
