"""Trainer / FusedAdamW / data staging on the GPU: reference-style loop equivalence, checkpoint formats, resume."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net(seed_from=None):
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd import model as M
    from oracle import harness
    cfg = GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    net = M.MMFN(cfg, DEV)
    net.load_state_dict((seed_from or harness.build_oracle("vec", dropout=0.0)).state_dict(), strict=True)
    return net, cfg


@pytest.fixture(scope="module")
def store(tmp_path_factory):
    from mmfn_amd import data as D
    from mmfn_amd.config import GlobalConfig
    from oracle import fixtures
    root = tmp_path_factory.mktemp("pro_train")
    samples = fixtures.synthetic_samples((5, 9, 3, 7), seed=3, radar_counts=(50, 81, 81, 20))
    for i, s in enumerate(samples):
        with open(root / ("%d.pkl" % i), "wb") as fd:
            pickle.dump(s, fd)
    return D.FrameStore(str(root), GlobalConfig(), "train")


def test_fused_epoch_equals_reference_style_loop(store):
    """Trainer.train(fused=True) == zero-grad / forward / F.l1_loss / backward / torch.optim.AdamW.step()."""
    from mmfn_amd import data as D
    from mmfn_amd.optim import FusedAdamW
    from mmfn_amd.trainer import Trainer
    from oracle import harness
    oracle = harness.build_oracle("vec", dropout=0.0)
    net_a, cfg = _net(oracle)
    net_b, _ = _net(oracle)
    loader = D.make_loader(store, batch_size=4, num_workers=0)  # one step per epoch
    ta, tb = Trainer(DEV, None), Trainer(DEV, None)
    opt_a, opt_b = FusedAdamW(net_a, lr=1e-4), torch.optim.AdamW(net_b.parameters(), lr=1e-4)
    la = ta.train(net_a, loader, cfg, opt_a)
    lb = tb.train(net_b, loader, cfg, opt_b, fused=False)
    assert ta.cur_iter == tb.cur_iter == 1 and ta.cur_epoch == tb.cur_epoch == 1
    assert abs(la - lb) <= 1e-6 * max(1.0, abs(lb))
    sa, sb = net_a.state_dict(), net_b.state_dict()
    for k in sa:
        if sa[k].dtype == torch.float32:
            assert (sa[k] - sb[k]).abs().max().item() <= 2e-6 * max(1.0, sb[k].abs().max().item()), k
    # second epoch: gradients at B=4 with these weights are ill-conditioned (test_e2e_gpu), so one-ulp differences
    # after step one already move individual updates; the epoch losses still have to agree
    la = ta.train(net_a, loader, cfg, opt_a)
    lb = tb.train(net_b, loader, cfg, opt_b, fused=False)
    assert abs(la - lb) <= 1e-4 * max(1.0, abs(lb)) and ta.train_loss[1] == la and ta.cur_iter == 2
    # validate() == the reference's eval loop written out (phase2_train_net.py:124-177) on the same weights
    va = ta.validate(net_a, loader, cfg)
    net_a.eval()
    ref, n = 0.0, 0
    with torch.no_grad():
        for batch in loader:
            args, gt = D.stage_batch(batch, DEV, cfg)
            ref += float(torch.nn.functional.l1_loss(net_a(*args), gt, reduction="none").mean())
            n += 1
    assert abs(va - ref / n) <= 1e-5 * max(1.0, abs(ref / n)) and ta.val_loss == [va]


def test_checkpoint_files_and_resume(store, tmp_path):
    from mmfn_amd import data as D
    from mmfn_amd.optim import FusedAdamW
    from mmfn_amd.trainer import Trainer
    from oracle import harness
    oracle = harness.build_oracle("vec", dropout=0.0)
    loader = D.make_loader(store, batch_size=2, num_workers=0)
    logdir = str(tmp_path / "log")

    net, cfg = _net(oracle)
    opt = FusedAdamW(net, lr=1e-4)
    tr = Trainer(DEV, logdir)
    tr.train(net, loader, cfg, opt)
    tr.validate(net, loader, cfg)
    assert tr.save(net, opt) is True
    for f in ("recent.log", "best_model.pth", "best_optim.pth", "model.pth", "recent_optim.pth"):
        assert os.path.isfile(os.path.join(logdir, f)), f
    table = json.load(open(os.path.join(logdir, "recent.log")))
    # the reference's six keys (phase2_train_net.py:165-176) plus this trainer's checkpoint stamps
    assert set(table) == {"epoch", "iter", "bestval", "bestval_epoch", "train_loss", "val_loss", "files"} and table["epoch"] == 1
    assert set(table["files"]) == {"model.pth", "recent_optim.pth", "best_model.pth", "best_optim.pth"}

    # the optimizer file is a torch.optim.AdamW state dict: torch loads it onto the same parameter list
    osd = torch.load(os.path.join(logdir, "best_optim.pth"))
    n_params = len(list(net.parameters()))
    assert len(osd["param_groups"]) == 1 and osd["param_groups"][0]["params"] == list(range(n_params))
    assert len(osd["state"]) == n_params - 21  # vec: raster-map stem + layer1 never get a gradient
    ref_opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    ref_opt.load_state_dict(osd)
    w = dict(net.named_parameters())["encoder.image_encoder.features.conv1.weight"]
    idx = [i for i, p in enumerate(net.parameters()) if p is w][0]
    assert ref_opt.state[w]["exp_avg"].shape == w.shape == osd["state"][idx]["exp_avg"].shape
    assert float(ref_opt.state[w]["step"]) == 2.0
    # ... and the weights file holds the reference's keys
    wsd = torch.load(os.path.join(logdir, "best_model.pth"))
    assert list(wsd.keys()) == list(oracle.state_dict().keys())

    # continue the original run for one more epoch; a fresh process resumes from disk and must land on the same weights
    tr.train(net, loader, cfg, opt)
    net2, _ = _net(oracle)
    opt2 = FusedAdamW(net2, lr=5e-4)
    tr2 = Trainer(DEV, logdir)
    assert tr2.resume(net2, opt2) is True
    assert tr2.cur_epoch == 1 and tr2.cur_iter == 2 and opt2.lr == 1e-4
    tr2.train(net2, loader, cfg, opt2)
    s1, s2 = net.state_dict(), net2.state_dict()
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
    assert tr2.train_loss == tr.train_loss


def test_graph_replay_epoch_equals_eager_epoch(store):
    """Trainer.train(graph=True): static-input hipGraph replay over changing batches == eager fused steps, bit for bit
    (dropout on: the counter RNG advances on the device inside the captured step)."""
    from mmfn_amd import data as D
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd import model as M
    from mmfn_amd.optim import FusedAdamW
    from mmfn_amd.trainer import Trainer
    from oracle import harness
    oracle = harness.build_oracle("vec", dropout=0.1)
    cfg = GlobalConfig()
    loader = D.make_loader(store, batch_size=1, num_workers=0)  # 4 batches with 5 / 9 / 3 / 7 lanes -> one 16-lane bucket
    nets = []
    for graph in (False, True):
        net = M.MMFN(cfg, DEV)
        net.load_state_dict(oracle.state_dict(), strict=True)
        tr = Trainer(DEV, None)
        opt = FusedAdamW(net, lr=1e-4)
        tr.train(net, loader, cfg, opt, graph=graph)
        tr.train(net, loader, cfg, opt, graph=graph)
        if graph:
            assert len(tr._static_steps) == 1  # one 16-lane bucket; ("eager" here would mean the capture fell back)
            assert not isinstance(next(iter(tr._static_steps.values())), str) or next(iter(tr._static_steps.values())) == "eager"
        nets.append((tr.train_loss, net.state_dict()))
    worst = max(((nets[0][1][k].double() - nets[1][1][k].double()).abs().max().item(), k) for k in nets[0][1])
    assert nets[0][0] == nets[1][0] and worst[0] == 0.0, (nets[0][0], nets[1][0], worst)


def test_epoch_from_packed_frames_equals_epoch_from_pickles(store, tmp_path):
    """data.PackedLoader (memory-mapped flat arrays gathered into pinned staging by a thread) feeds Trainer.train exactly what
    the DataLoader over the phase-1 pickles feeds it: two epochs, dropout on, graph replay - losses and every weight bit-equal."""
    from mmfn_amd import data as D
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd import model as M
    from mmfn_amd.optim import FusedAdamW
    from mmfn_amd.trainer import Trainer
    from oracle import harness
    oracle = harness.build_oracle("vec", dropout=0.1)
    cfg = GlobalConfig()
    packed = D.PackedFrames(D.pack_frames(store, str(tmp_path / "packed")))
    loaders = (D.make_loader(store, batch_size=2, num_workers=0), D.PackedLoader(packed, batch_size=2))
    out = []
    for loader in loaders:
        net = M.MMFN(cfg, DEV)
        net.load_state_dict(oracle.state_dict(), strict=True)
        tr = Trainer(DEV, None)
        opt = FusedAdamW(net, lr=1e-4)
        tr.train(net, loader, cfg, opt)
        tr.train(net, loader, cfg, opt)
        out.append((tr.train_loss, net.state_dict()))
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k


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


def test_data_parallel_epoch_on_the_single_graph_transport_equals_plain_epoch(store, monkeypatch):
    """Trainer.train(dp=parallel.connect(...)): with the C-ABI RCCL transport the training loop's step - forward, backward, every
    gradient-bucket all-reduce, AdamW - is ONE captured hipGraph per batch shape.  One GPU: a 1-rank communicator (the sum over
    one rank is the identity), so two epochs must leave exactly the weights of the plain single-GPU epochs; the transport
    must really be the C ABI and the replayed step really a single graph."""
    from mmfn_amd import data as D
    from mmfn_amd import parallel as P
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd import model as M
    from mmfn_amd.optim import FusedAdamW
    from mmfn_amd.trainer import Trainer
    from oracle import harness
    oracle = harness.build_oracle("vec", dropout=0.1)
    cfg = GlobalConfig()
    loader = D.make_loader(store, batch_size=1, num_workers=0)
    monkeypatch.delenv("MMFN_BENCH_SINGLE_DEVICE", raising=False)
    monkeypatch.setattr(P.DataParallel, "broadcast_parameters", lambda self, src=0: None)   # (nothing to broadcast to)
    out = []
    for use_dp in (False, True):
        net = M.MMFN(cfg, DEV)
        net.load_state_dict(oracle.state_dict(), strict=True)
        dp = None
        if use_dp:
            import mmfn_amd.comm as C
            monkeypatch.setattr(C, "all_ranks_agree", lambda dist, dev, ok: ok)
            dp, note = P.connect(net, _OneRank, transport="capi")
            assert dp.comm is not None and note is None and dp.n_buckets() >= 17
        tr = Trainer(DEV, None)
        opt = FusedAdamW(net, lr=1e-4)
        tr.train(net, loader, cfg, opt, dp=dp)
        tr.train(net, loader, cfg, opt, dp=dp)
        if use_dp:
            step = next(iter(tr._static_steps.values()))
            assert not isinstance(step, str) and step.seg.single_graph and step.seg.recorder.n_graphs == 1
            del step
            tr._static_steps.clear()   # the captures go before the communicator they recorded (RcclComm.destroy)
            dp.comm.destroy()
        out.append((tr.train_loss, net.state_dict()))
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k


def test_graph_replayed_validation_equals_eager_validation(store):
    """Trainer.validate(graph=True): the eval forward captured per input shape and replayed over changing batches gives the same
    validation loss as eager launches, before and after the weights change (the capture reads the weights at replay time)."""
    from mmfn_amd import data as D
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd import model as M
    from mmfn_amd.optim import FusedAdamW
    from mmfn_amd.parallel import StaticEvalStep
    from mmfn_amd.trainer import Trainer
    from oracle import harness
    oracle = harness.build_oracle("vec", dropout=0.0)
    cfg = GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    loader = D.make_loader(store, batch_size=1, num_workers=0)  # 4 batches with 5 / 9 / 3 / 7 lanes -> one 16-lane bucket
    net = M.MMFN(cfg, DEV)
    net.load_state_dict(oracle.state_dict(), strict=True)
    tr_g, tr_e = Trainer(DEV, None), Trainer(DEV, None)
    opt = FusedAdamW(net, lr=1e-3)
    for _ in range(2):
        vg = tr_g.validate(net, loader, cfg, graph=True)
        ve = tr_e.validate(net, loader, cfg, graph=False)
        assert vg == ve, (vg, ve)
        tr_g.train(net, loader, cfg, opt, graph=False)     # move the weights, then validate again through the same capture
    assert len(tr_g._static_evals) == 1 and isinstance(next(iter(tr_g._static_evals.values())), StaticEvalStep)
    assert tr_g.val_loss[0] != tr_g.val_loss[1]


def test_param_groups_match_torch_adamw_and_round_trip(store, tmp_path):
    """The reference's decay / no-decay groups (model_vec.py:179-209) through FusedAdamW == torch.optim.AdamW with the same
    groups; the multi-group optimizer state file loads into torch's optimizer and back."""
    from mmfn_amd import data as D
    from mmfn_amd.optim import FusedAdamW, configure_optimizers
    from mmfn_amd.trainer import Trainer
    from oracle import harness
    oracle = harness.build_oracle("vec", dropout=0.0)
    net_a, cfg = _net(oracle)
    net_b, _ = _net(oracle)
    ga, gb = configure_optimizers(net_a, weight_decay=0.05), configure_optimizers(net_b, weight_decay=0.05)
    assert len(ga[0]["params"]) + len(ga[1]["params"]) == len(list(net_a.parameters()))
    opt_a = FusedAdamW(net_a, lr=3e-4, param_groups=ga)
    opt_b = torch.optim.AdamW(gb, lr=3e-4)
    loader = D.make_loader(store, batch_size=4, num_workers=0)
    ta, tb = Trainer(DEV, None), Trainer(DEV, None)
    ta.train(net_a, loader, cfg, opt_a, graph=False)
    tb.train(net_b, loader, cfg, opt_b, fused=False)
    sa, sb = net_a.state_dict(), net_b.state_dict()
    for k in sa:
        if sa[k].dtype == torch.float32:
            assert (sa[k] - sb[k]).abs().max().item() <= 2e-6 * max(1.0, sb[k].abs().max().item()), k
    # weight decay really differs between the groups: a no-decay tensor with zero gradient history would be untouched,
    # here simply compare against a single-group run
    net_c, _ = _net(oracle)
    opt_c = FusedAdamW(net_c, lr=3e-4, weight_decay=0.05)
    Trainer(DEV, None).train(net_c, loader, cfg, opt_c, graph=False)
    bias = "encoder.transformer4.blocks.0.mlp.0.bias"
    wname = "encoder.transformer4.blocks.0.mlp.0.weight"
    sc = net_c.state_dict()
    assert torch.equal(sa[wname], sc[wname]) and not torch.equal(sa[bias], sc[bias])
    # torch-format state with two groups: ids run group by group
    osd = opt_a.state_dict()
    assert [len(g["params"]) for g in osd["param_groups"]] == [len(ga[0]["params"]), len(ga[1]["params"])]
    assert osd["param_groups"][0]["weight_decay"] == 0.05 and osd["param_groups"][1]["weight_decay"] == 0.0
    ref_opt = torch.optim.AdamW(configure_optimizers(net_a, weight_decay=0.05), lr=1.0)
    ref_opt.load_state_dict(osd)
    p0 = ga[0]["params"][0]
    assert torch.equal(ref_opt.state[p0]["exp_avg"], osd["state"][0]["exp_avg"]) and ref_opt.param_groups[0]["lr"] == 3e-4
    net_d, _ = _net(oracle)
    opt_d = FusedAdamW(net_d, lr=1.0, param_groups=configure_optimizers(net_d))
    opt_d.load_state_dict(ref_opt.state_dict())
    assert opt_d.param_groups[0]["weight_decay"] == 0.05 and opt_d.lr == 3e-4
    assert torch.equal(net_d._layout.exp_avg, net_a._layout.exp_avg) and torch.equal(net_d._layout.exp_avg_sq, net_a._layout.exp_avg_sq)


def test_lr_schedule_does_not_recapture_and_matches_eager(store):
    """A learning rate that changes every step: the captured step keeps replaying (hyper-parameters are read from device
    memory), the capture cache stays at one entry, and the result equals the eager run bit for bit."""
    from mmfn_amd import data as D
    from mmfn_amd.optim import FusedAdamW
    from mmfn_amd.trainer import Trainer
    from oracle import harness
    oracle = harness.build_oracle("vec", dropout=0.0)
    loader = D.make_loader(store, batch_size=1, num_workers=0)
    outs = []
    for graph in (False, True):
        net, cfg = _net(oracle)
        opt = FusedAdamW(net, lr=1e-4)
        tr = Trainer(DEV, None)
        for epoch in range(3):
            opt.param_groups[0]["lr"] = 1e-4 * (0.5 ** epoch)
            tr.train(net, loader, cfg, opt, graph=graph)
        if graph:
            assert len(tr._static_steps) == 1 and not isinstance(next(iter(tr._static_steps.values())), str)
        outs.append(net.state_dict())
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_resume_without_validation_and_atomic_files(store, tmp_path):
    """No validation run -> save() writes no best_* pair; resume(which="best") falls back to the recent pair instead of
    raising.  No temporary files are left behind."""
    from mmfn_amd import data as D
    from mmfn_amd.optim import FusedAdamW
    from mmfn_amd.trainer import Trainer
    from oracle import harness
    oracle = harness.build_oracle("vec", dropout=0.0)
    loader = D.make_loader(store, batch_size=2, num_workers=0)
    logdir = str(tmp_path / "log_noval")
    net, cfg = _net(oracle)
    opt = FusedAdamW(net, lr=1e-4)
    tr = Trainer(DEV, logdir)
    tr.train(net, loader, cfg, opt)
    assert tr.save(net, opt) is False
    assert sorted(os.listdir(logdir)) == ["model.pth", "recent.log", "recent_optim.pth"]
    net2, _ = _net(oracle)
    opt2 = FusedAdamW(net2, lr=1.0)
    tr2 = Trainer(DEV, logdir)
    assert tr2.resume(net2, opt2) is True and tr2.cur_epoch == 1 and opt2.lr == 1e-4
    s1, s2 = net.state_dict(), net2.state_dict()
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
