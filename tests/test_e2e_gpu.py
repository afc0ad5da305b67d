"""End-to-end parity of the HIP MMFN against the CPU oracle on identical weights and batches."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(variant="vec", B=2, lanes=None, dropout=0.0):
    lanes = (9 if variant != "img" else 4) if lanes is None else lanes  # as oracle/make_golden.py
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd import model as M
    from oracle import fixtures, harness
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    oracle = harness.build_oracle(variant, dropout=dropout)
    cfg = GlobalConfig(embd_pdrop=dropout, attn_pdrop=dropout, resid_pdrop=dropout)
    cls = {"vec": M.MMFN, "img": M.MMFNImg, "rad": M.MMFNRad}[variant]
    net = cls(cfg, DEV)
    net.load_state_dict(oracle.state_dict(), strict=True)
    batch = fixtures.synthetic_batch(B, variant, seed=42, lanes=lanes)
    args = harness.forward_args(batch, variant)
    return oracle, net, batch, args


def _dev_args(args):
    to = lambda t: t.to(DEV)
    img, lid, maps, vm, radar, adj, tp, vel = args
    vmd = [[to(vm[0][0])], [to(vm[1][0])], vm[2]]
    return ([to(img[0])], [to(lid[0])], [to(maps[0])], vmd, [to(radar[0])], [to(adj[0])], to(tp), to(vel))


@pytest.mark.parametrize("variant", ["vec", "img", "rad"])
def test_eval_forward_matches_oracle(variant):
    from oracle import harness
    oracle, net, batch, args = _setup(variant)
    harness.calibrate_bn(oracle, args)
    net.load_state_dict(oracle.state_dict(), strict=True)  # calibrated running stats
    net.eval()
    with torch.no_grad():
        ref = oracle(*args)
        got = net(*_dev_args(args)).cpu()
    err = (got - ref).abs().max().item()
    assert err <= 1e-4, "waypoint max abs err %g" % err


@pytest.mark.parametrize("variant", ["vec", "rad", "img"])
def test_golden_eval_waypoints(golden_dir, variant):
    """Same check against the committed vectors produced by the reference itself."""
    from oracle import harness
    oracle, net, batch, args = _setup(variant)
    g = np.load(os.path.join(golden_dir, "mmfn_%s_b2.npz" % variant))
    harness.calibrate_bn(oracle, args)
    net.load_state_dict(oracle.state_dict(), strict=True)
    net.eval()
    with torch.no_grad():
        got = net(*_dev_args(args)).cpu().numpy()
    assert np.abs(got - g["eval_pred_wp"]).max() <= 1e-4
    if variant == "img":   # the image agent has no vector map: its batch-1 call is the plain forward signature
        if "eval_pred_wp_b1_agent" in g.files:
            with torch.no_grad():
                d = _dev_args(args)
                got1 = net([d[0][0][:1]], [d[1][0][:1]], [d[2][0][:1]], None, None, None, d[6][:1], d[7][:1]).cpu().numpy()
            assert np.abs(got1 - g["eval_pred_wp_b1_agent"]).max() <= 1e-4
        return
    # agent-style call: batch 1, vectormap lane count passed as tensors (mmfn_vectornet.py:287-297)
    one = [[batch["lane"][:1][None].to(DEV)], [batch["lane_num"][:1].int().view(1, 1).to(DEV)],
           batch["lane_num"][:1].int().view(1, 1).to(DEV)]
    with torch.no_grad():
        got1 = net([args[0][0][:1].to(DEV)], [args[1][0][:1].to(DEV)], None, one, [batch["radar"][:1].to(DEV)],
                   [batch["radar_adj"][:1].to(DEV)], batch["target_point"][:1].to(DEV),
                   batch["velocity"][:1].to(DEV)).cpu().numpy()
    assert np.abs(got1 - g["eval_pred_wp_b1_agent"]).max() <= 1e-4


def _to64(a):
    if torch.is_tensor(a):
        return a.double() if a.is_floating_point() else a
    if isinstance(a, (list, tuple)):
        return type(a)(_to64(x) for x in a)
    return a


@pytest.mark.parametrize("variant", ["vec", "img", "rad"])
def test_train_step_matches_oracle(variant, golden_dir):
    """loss, every parameter gradient, BN running stats and the AdamW update of one step.

    Gradients: with batch 2 the backward through 85 train-mode BatchNorms is ill-conditioned — the
    fp32 CPU oracle itself deviates from an fp64 evaluation of the same graph by up to ~30 % on
    some tensors.  So the HIP gradients are judged against the fp64 oracle, with the fp32 oracle's
    own error on the same tensor as the yardstick."""
    import copy
    from oracle import harness
    oracle, net, batch, args = _setup(variant)
    init_sd = {k: v.detach().clone() for k, v in oracle.state_dict().items()}
    o64 = copy.deepcopy(oracle).double()
    _, loss64, g64 = harness.train_step(o64, _to64(args), batch["gt_wp"].double())
    pred_ref, loss_ref, grads_ref = harness.train_step(oracle, args, batch["gt_wp"])
    net.train()
    for p in net.parameters():
        p.grad = None
    pred = net(*_dev_args(args))
    loss = torch.nn.functional.l1_loss(pred, batch["gt_wp"].to(DEV), reduction="none").mean()
    loss.backward()
    assert (pred.detach().cpu() - pred_ref).abs().max().item() <= 1e-4
    assert abs(loss.item() - loss_ref.item()) <= 1e-4, (loss.item(), loss_ref.item())
    assert abs(loss.item() - loss64.item()) <= 2e-5, (loss.item(), loss64.item())
    g = np.load(os.path.join(golden_dir, "mmfn_%s_b2.npz" % variant))
    assert abs(loss.item() - float(g["train_loss"])) <= 1e-4
    gmax = max(t.norm().item() for t in g64.values() if t is not None)
    # typical relative error of the fp32 oracle against fp64: the yardstick for tensors where the oracle happened to land
    # unusually close (the error ratio is long-tailed in both directions)
    rel_cpu = sorted((grads_ref[k].double() - t).norm().item() / t.norm().item() for k, t in g64.items()
                     if t is not None and t.norm().item() > 1e-6 * gmax)
    med_cpu = rel_cpu[len(rel_cpu) // 2]
    bad, ratios = [], []
    for name, p in net.named_parameters():
        t = g64[name]
        if t is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        n = t.norm().item()
        e_gpu = (p.grad.detach().cpu().double() - t).norm().item()
        e_cpu = (grads_ref[name].double() - t).norm().item()
        if n > 1e-6 * gmax:
            ratios.append(e_gpu / max(e_cpu, 1e-12 * gmax))
        if e_gpu > 12.0 * max(e_cpu, med_cpu * n) + 2e-4 * n + 1e-8 * gmax:
            bad.append((name, e_gpu, e_cpu, n))
    assert not bad, "gradient error (name, |gpu-f64|, |cpu32-f64|, |f64|): %s" % bad[:8]
    ratios.sort()
    assert ratios[len(ratios) // 2] <= 2.5, "median HIP/CPU-fp32 gradient error ratio %g" % ratios[len(ratios) // 2]
    # the tail of this statistic is chaotic: a last-ulp change anywhere in the forward (e.g. a different but equally
    # accurate summation order in the attention kernels) moves p95 between ~3.5 and ~6 on this batch, and other seeds
    # give 4..17 for every kernel generation tried (tools/grad_ratio.py); the per-tensor bound above is the hard check
    assert ratios[int(len(ratios) * 0.95)] <= 8.0, "95th percentile gradient error ratio %g" % ratios[int(len(ratios) * 0.95)]
    # ---- the reference-generated vectors (tests/golden, oracle/make_golden.py): per-parameter gradient norms / first
    # elements and the stage taps localise a drift to a layer.  The golden gradients are the
    # fp32 reference's, so they carry that run's own fp32 error e_cpu (measured above against fp64) on top of ours.
    names = [str(n) for n in g["param_names"]]
    params = dict(net.named_parameters())
    bad = []
    for i, name in enumerate(names):
        p = params[name]
        if bool(g["grad_none"][i]):
            assert p.grad is None, name
            continue
        t = g64[name]
        n = t.norm().item()
        e_cpu = (grads_ref[name].double() - t).norm().item()
        tol = 13.0 * max(e_cpu, med_cpu * n) + 2e-4 * n + 1e-8 * gmax
        gn = p.grad.detach().double().norm().item()
        head = p.grad.detach().flatten()[:8].cpu().double().numpy()
        m = head.size
        if abs(gn - float(g["grad_norm"][i])) > tol or np.abs(head - g["grad_head"][i][:m].astype(np.float64)).max() > tol:
            bad.append((name, gn, float(g["grad_norm"][i]), tol))
    assert not bad, "gradient vs reference-generated golden vectors (name, |hip|, |golden|, tol): %s" % bad[:8]
    eng = net._engine_for()
    taps = {"fused": eng.taps["fused"]}
    for s_ in range(4):  # image tokens of GPT s: [B, 64, C] rows == the reference's NCHW [B, C, 8, 8] image output
        taps["gpt%d_img" % (s_ + 1)] = eng.taps["gpt%d" % (s_ + 1)][:, :64].permute(0, 2, 1)
    if variant != "img":
        taps["vectornet"] = eng.taps["stage1"][2].permute(0, 3, 1, 2)  # NHWC map features -> "b n d a"
    for k, v in taps.items():
        v = v.contiguous().double().cpu()
        ref_abs = float(g["tap_%s_abs" % k])
        assert abs(v.sum().item() - float(g["tap_%s_sum" % k])) <= 1e-4 * ref_abs + 1e-6, k
        assert abs(v.abs().sum().item() - ref_abs) <= 1e-4 * ref_abs + 1e-6, k
        hd = g["tap_%s_head" % k].astype(np.float64)
        # element-wise: two fp32 evaluations of 4 fusion scales at batch 2 agree to a few 1e-4 on single activations
        assert np.abs(v.flatten()[:16].numpy() - hd).max() <= 1e-3 * max(1.0, np.abs(hd).max()), k
    # ---- optimizer: torch AdamW on the views == what the reference loop does.  The first Adam step moves every weight
    # by lr * g / (|g| + eps) ~ +-lr, so a meaningful check is elementwise on the UPDATE, restricted to the elements whose
    # gradient sign fp32 arithmetic determines at all (|g| well above the fp32 oracle's own error against fp64).
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    opt.step()
    ref_sd = oracle.state_dict()   # the oracle after ITS AdamW step (harness.train_step)
    got_sd = net.state_dict()
    checked = total = 0
    for k, v in ref_sd.items():
        if v.dtype != torch.float32:
            assert int(got_sd[k].item()) == int(v.item()), k
            continue
        if "running" in k:
            # batch-2 statistics: fp32 noise through 30 layers - a layer4 BatchNorm sees 128 values per channel, and every
            # Winograd convolution (all 3x3 stride-1 layers from 64 channels on) differs from the direct form by <= 2e-5
            # relative.  Measured worst case 1.1e-3 of the largest running variance; the batch-32 test holds 1e-3.
            d = (got_sd[k].cpu() - v).abs().max().item()
            assert d <= 2e-3 * max(1.0, v.abs().max().item()), (k, d)
            continue
        if g64.get(k) is None:
            assert torch.equal(got_sd[k].cpu(), before[k].cpu()), k   # no gradient -> untouched (torch semantics)
            continue
        t = g64[k]
        # elements whose gradient sign is determined on BOTH sides (gradient accuracy itself is judged above; this block
        # checks the update rule: decoupled decay, bias corrections, sign and size of the step)
        ghip = params[k].grad.detach().cpu().double()
        sure = t.abs() > 10.0 * torch.maximum((grads_ref[k].double() - t).abs(), (ghip - t).abs()) + 1e-7 * gmax
        upd_hip = (got_sd[k].cpu().double() - before[k].cpu().double())
        # oracle's update, reconstructed from its post-step weights and the closed-form initial fill
        upd_ref = (v.double() - init_sd[k].double())
        total += t.numel()
        checked += int(sure.sum())
        if sure.any():
            dd = (upd_hip - upd_ref).abs() * sure
            d = dd.max().item()
            if d > 2e-6:   # lr = 1e-4: a wrong sign is 2e-4, a missing update 1e-4
                i = int(dd.flatten().argmax())
                raise AssertionError((k, d, "elem %d: g64 %g g32 %g ghip %g upd_hip %g upd_ref %g"
                                      % (i, t.flatten()[i], grads_ref[k].flatten()[i], ghip.flatten()[i], upd_hip.flatten()[i],
                                         upd_ref.flatten()[i])))
    assert checked >= 0.1 * total, (checked, total)   # ~20 % of the elements have a gradient sign both fp32 runs determine


def test_fused_train_step_equals_autograd_path():
    """engine.train_step (fused loss + flat AdamW) == autograd bridge + torch AdamW."""
    oracle, net_a, batch, args = _setup("vec")
    _, net_b, _, _ = _setup("vec")
    dargs = _dev_args(args)
    net_a.train()
    pred = net_a(*dargs)
    loss_a = torch.nn.functional.l1_loss(pred, batch["gt_wp"].to(DEV), reduction="none").mean()
    loss_a.backward()
    torch.optim.AdamW(net_a.parameters(), lr=1e-4).step()
    net_b.train()
    inp = net_b._pack(*dargs)
    loss_b = net_b.train_step(inp, batch["gt_wp"].to(DEV))
    assert abs(loss_a.item() - loss_b.item()) <= 1e-6
    sa, sb = net_a.state_dict(), net_b.state_dict()
    for k in sa:
        if sa[k].dtype == torch.float32:
            assert (sa[k] - sb[k]).abs().max().item() <= 1e-6, k


def test_backward_accumulates_into_existing_gradients_like_torch():
    """loss.backward() twice without zeroing: .grad holds the SUM (torch.autograd semantics the reference's loop relies on
    implicitly, phase2_train_net.py:60 zero_grad / :106 backward); after p.grad = None / zero_grad() the next backward starts
    afresh.  The explicit HIP backward overwrites its flat buffer, so the bridge parks and re-adds what was attached."""
    oracle, net, batch, args = _setup("vec")
    dargs = _dev_args(args)
    gt = batch["gt_wp"].to(DEV)
    net.train()
    loss = lambda: torch.nn.functional.l1_loss(net(*dargs), gt, reduction="none").mean()
    loss().backward()
    once = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    assert len(once) > 800
    loss().backward()                                   # second backward, nothing zeroed: 2 x the gradient, exactly
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, 2.0 * once[n]), n
    net.zero_grad()                                     # set_to_none: a fresh start
    assert all(p.grad is None for p in net.parameters())
    loss().backward()
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, once[n]), n
    # the oracle does the same
    for p in oracle.parameters():
        p.grad = None
    oracle.train()
    for _ in range(2):
        torch.nn.functional.l1_loss(oracle(*args), batch["gt_wp"], reduction="none").mean().backward()
    ref = dict(oracle.named_parameters())["join.0.weight"].grad
    net.zero_grad()
    loss().backward(); loss().backward()
    got = dict(net.named_parameters())["join.0.weight"].grad.cpu()
    assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()


def test_raw_sensor_ingest_path_equals_tensor_path():
    """u8 camera frames + XYZI points through the GPU ingest/splat kernels == preprocessed tensors."""
    oracle, net, batch, args = _setup("vec")
    net.eval()
    dargs = _dev_args(args)
    with torch.no_grad():
        ref = net(*dargs)
        inp = net._pack(*dargs)
        raw = dict(inp)
        del raw["image"], raw["lidar"]
        raw["rgb_u8"] = batch["rgb_u8"].to(DEV)
        raw["lidar_pts"] = batch["lidar_pts"].to(DEV)
        pred, _ = net._engine_for().forward(raw, False, None)
    assert torch.equal(pred, ref)


def test_segmented_graph_step_equals_eager_steps():
    """parallel.GraphedStep (one hipGraph on one GPU; cut at the gradient-bucket boundaries under torch.distributed data parallelism) replays to exactly the
    parameters the eager train_step produces, dropout included (counter RNG advances on the device)."""
    from mmfn_amd.parallel import GraphedStep
    _, net_a, batch, args = _setup("vec", dropout=0.1)
    _, net_b, _, _ = _setup("vec", dropout=0.1)
    dargs = _dev_args(args)
    gt = batch["gt_wp"].to(DEV)
    net_a.train(), net_b.train()
    inp_a, inp_b = net_a._pack(*dargs), net_b._pack(*dargs)
    for _ in range(3):
        loss_a = net_a.train_step(inp_a, gt)
    step = GraphedStep(net_b._engine_for(), None, inp_b, gt, warm=1)
    for _ in range(2):
        loss_b = step()
    torch.cuda.synchronize()
    assert loss_a.item() == loss_b.item()
    sa, sb = net_a.state_dict(), net_b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_failed_capture_leaves_the_stream_usable(monkeypatch):
    """a capture that raises (here: inside a branch lane, and inside the main graph) must END the
    stream capture before the exception travels on - otherwise the callers' eager fallback dies in synchronize()."""
    from mmfn_amd import engine as E
    from mmfn_amd.parallel import GraphedStep
    _, net, batch, args = _setup("vec", dropout=0.1)
    dargs = _dev_args(args)
    gt = batch["gt_wp"].to(DEV)
    net.train()
    inp = net._pack(*dargs)
    net.train_step(inp, gt)
    torch.cuda.synchronize()
    real_vec, real_opt = E.VectorNet.bwd, E.Engine.optimizer_step

    def boom(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("injected capture failure")
        return real_vec(*a, **k)

    monkeypatch.setattr(E.VectorNet, "bwd", boom)    # fails inside a branch lane: a side stream forked into the capture
    with pytest.raises(RuntimeError, match="injected"):
        GraphedStep(net._engine_for(), None, inp, gt, warm=0)
    monkeypatch.setattr(E.VectorNet, "bwd", real_vec)
    assert not torch.cuda.is_current_stream_capturing()
    torch.cuda.synchronize()                         # illegal while any stream of this thread is still capturing
    loss = net.train_step(inp, gt)                   # the eager fallback the trainer / bench.py take
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all()

    def boom_opt(self, *a, **k):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("injected capture failure")
        return real_opt(self, *a, **k)

    monkeypatch.setattr(E.Engine, "optimizer_step", boom_opt)
    with pytest.raises(RuntimeError, match="injected"):
        GraphedStep(net._engine_for(), None, inp, gt, warm=0)
    monkeypatch.setattr(E.Engine, "optimizer_step", real_opt)
    torch.cuda.synchronize()
    step = GraphedStep(net._engine_for(), None, inp, gt, warm=0)   # and a later capture works
    step()
    torch.cuda.synchronize()
    assert torch.isfinite(step.loss).all()


@pytest.mark.parametrize("variant", ["vec", "img", "rad"])
def test_driving_session_reproduces_the_agents_batch1_vectors(variant, golden_dir):
    """DrivingSession for the three agents (e2e_agent/mmfn_vectornet.py, mmfn_imgnet.py, mmfn_radar.py): raw u8 frame, raw sweep
    (the session flips y as the agents do), lanes / bird's-eye raster / raw radar returns in, against the waypoints the
    REFERENCE produced for the same sample in its agent-style batch-1 call (tests/golden `eval_pred_wp_b1_agent`) - with the
    BatchNorms folded into the filters and the tick replayed as one hipGraph, and unfolded / eager."""
    from mmfn_amd.inference import DrivingSession
    from oracle import harness
    g = np.load(os.path.join(golden_dir, "mmfn_%s_b2.npz" % variant))
    oracle, net, batch, args = _setup(variant, B=2)
    harness.calibrate_bn(oracle, args)       # as oracle/make_golden.py: running statistics := statistics of the batch-2 call
    net.load_state_dict(oracle.state_dict(), strict=True)
    rgb = batch["rgb_u8"][0].numpy()
    pts = batch["lidar_pts"][0].numpy().copy()
    pts[:, 1] *= -1                          # the fixture's sweep is in the ego frame already; the agent's y flip undoes this
    kw = {}
    lanes = None
    if variant == "img":
        kw["map_image"] = batch["map_u8"][0].numpy().transpose(1, 2, 0)     # HWC, as the agent holds it
    else:
        lanes = batch["lane"][0, :int(batch["lane_num"][0])].numpy()
    if variant == "rad":
        kw["radar"] = batch["radar"][0].numpy()                              # 81 returns: radar_to_size keeps them as they are
    tp, speed = batch["target_point"][0].tolist(), float(batch["velocity"][0])
    ref = g["eval_pred_wp_b1_agent"]
    outs = []
    for fold, graph in ((True, True), (False, False)):
        sess = DrivingSession(net, max_points=1 << 15, max_lanes=32, use_graph=graph, fold_batchnorm=fold)
        got = sess.predict(rgb, pts, lanes, tp, speed, merge_previous_sweep=False, **kw).numpy()
        assert np.abs(got - ref).max() <= 1e-4, (variant, fold, np.abs(got - ref).max())
        outs.append(got)
        got2 = sess.predict(rgb, pts, lanes, tp, speed, merge_previous_sweep=False, **kw).numpy()   # a second tick over the same buffers
        assert np.array_equal(got, got2)
    out = sess.run_step(rgb, pts, lanes, tp, speed, **kw)
    assert -1.0 <= out["steer"] <= 1.0 and 0.0 <= out["throttle"] <= 0.75
    if variant != "vec":
        with pytest.raises(ValueError):
            sess.predict(rgb, pts, lanes, tp, speed)      # the raster / the radar returns are missing


def test_driving_session_equals_reference_pipeline(golden_dir):
    """Batch-1 closed-loop entry (raw u8 frame + XYZI sweep + ragged lanes, hipGraph) == the agent's sequence:
    oracle crop / y-flip / histogram on the CPU, then the oracle network, then control_pid."""
    from mmfn_amd.inference import DrivingSession
    from oracle import harness, preprocess
    oracle, net, batch, args = _setup("vec", B=2)
    harness.calibrate_bn(oracle, args)
    net.load_state_dict(oracle.state_dict(), strict=True)
    oracle.eval()
    sess = DrivingSession(net, max_points=1 << 15, max_lanes=32)
    eager = DrivingSession(net, max_points=1 << 15, max_lanes=32, use_graph=False)
    rng = np.random.RandomState(0)
    prev = None
    for tick in range(3):
        rgb = rng.randint(0, 256, (300, 400, 3)).astype(np.uint8)
        pts = np.stack([rng.uniform(-20, 20, 6000), rng.uniform(-12, 28, 6000), rng.uniform(-3, 1, 6000),
                        rng.uniform(0, 1, 6000)], 1).astype(np.float32)
        L = [7, 3, 12][tick]
        lanes = rng.randn(L, 10, 5).astype(np.float32)
        tp, speed = (float(rng.randn() * 10), float(rng.randn() * 10)), float(rng.uniform(0, 8))
        got = sess.predict(rgb, pts, lanes, tp, speed)
        assert torch.equal(got, eager.predict(rgb, pts, lanes, tp, speed))
        # reference-side pipeline on the CPU
        sweep = pts if prev is None else np.append(pts, prev, axis=0)
        prev = pts
        flipped = sweep[:, :3].astype(np.float64).copy()
        flipped[:, 1] *= -1
        bev = torch.from_numpy(preprocess.lidar_histogram(flipped))[None]
        img = torch.from_numpy(preprocess.crop_chw(rgb).astype(np.float32))[None]
        vm = [[torch.from_numpy(lanes)[None]], [torch.tensor([float(L)])], L]
        with torch.no_grad():
            ref = oracle([img], [bev], None, vm, None, None, torch.tensor([tp], dtype=torch.float32), torch.tensor([speed]))
        assert (got - ref).abs().max().item() <= 1e-4, tick
    out = sess.run_step(rgb, pts, lanes, tp, speed)
    assert set(out) == {"steer", "throttle", "brake", "pred_wp", "pid"} and -1.0 <= out["steer"] <= 1.0
    with pytest.raises(ValueError):
        sess.predict(rgb[:100], pts, lanes, tp, speed)


def test_bf16_operand_mode_tracks_the_fp32_path():
    """GlobalConfig(gemm_dtype="bf16"): Linear GEMMs and direct convolutions take bf16 MFMA operands (fp32 accumulation,
    activations, master weights).  Batch-2 smoke check with the closed-form test weights: waypoints and loss stay within
    bf16's rounding of the fp32 path and the backward runs.  The gradient criterion (per-stage cosine against torch's own
    autocast as the yardstick, reference-style init, batch 8 and 32) is tests/test_parity_benchsize_gpu.py::
    test_bf16_mode_tracks_fp32_at_least_as_well_as_torch_autocast."""
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd import model as M
    from oracle import harness
    oracle, net32, batch, args = _setup("vec")
    cfg16 = GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, gemm_dtype="bf16")
    net16 = M.MMFN(cfg16, DEV)
    net16.load_state_dict(oracle.state_dict(), strict=True)
    dargs = _dev_args(args)
    gt = batch["gt_wp"].to(DEV)
    losses, grads = [], []
    for net in (net32, net16):
        net.train()
        for p in net.parameters():
            p.grad = None
        loss = torch.nn.functional.l1_loss(net(*dargs), gt, reduction="none").mean()
        loss.backward()
        losses.append(loss.item())
        grads.append(net._layout.grads[:net._layout.tail].clone())
    assert abs(losses[0] - losses[1]) <= 2e-2 * abs(losses[0]), losses
    assert torch.isfinite(grads[1]).all() and grads[1].abs().max().item() > 0
    harness.calibrate_bn(oracle, args)
    for net in (net32, net16):
        net.load_state_dict(oracle.state_dict(), strict=True)
        net.eval()
    with torch.no_grad():
        a, b = net32(*dargs), net16(*dargs)
    assert (a - b).abs().max().item() <= 2e-2 * max(1.0, a.abs().max().item()), (a - b).abs().max().item()


def test_folded_batchnorm_session_matches_unfolded_and_follows_weight_changes():
    """DrivingSession folds the eval-mode BatchNorms into the convolution filters (conv + shift + skip + ReLU in one launch,
    MMFN_EPI_RELU_LAST): same waypoints as the unfolded eval forward, and refresh() picks up changed weights / running statistics
    without a new capture."""
    from mmfn_amd.inference import DrivingSession
    from oracle import harness
    oracle, net, batch, args = _setup("vec", B=2)
    harness.calibrate_bn(oracle, args)
    net.load_state_dict(oracle.state_dict(), strict=True)
    folded = DrivingSession(net, max_points=1 << 14, max_lanes=16)
    plain = DrivingSession(net, max_points=1 << 14, max_lanes=16, fold_batchnorm=False)
    rng = np.random.RandomState(1)
    rgb = rng.randint(0, 256, (300, 400, 3)).astype(np.uint8)
    pts = np.stack([rng.uniform(-20, 20, 5000), rng.uniform(-12, 28, 5000), rng.uniform(-3, 1, 5000), rng.uniform(0, 1, 5000)], 1).astype(np.float32)
    lanes = rng.randn(6, 10, 5).astype(np.float32)

    def both():
        a = folded.predict(rgb, pts, lanes, (2.0, 15.0), 3.0, merge_previous_sweep=False)
        b = plain.predict(rgb, pts, lanes, (2.0, 15.0), 3.0, merge_previous_sweep=False)
        return a, b

    a0, b0 = both()
    assert (a0 - b0).abs().max().item() <= 2e-5
    with torch.no_grad():   # change filters and running statistics in place
        for name, p in net.named_parameters():
            if "image_encoder" in name and name.endswith("conv1.weight"):
                p.mul_(1.05)
        for name, b in net.named_buffers():
            if name.endswith("running_var"):
                b.mul_(1.1)
    a1, b1 = both()
    assert (b1 - b0).abs().max().item() > 1e-4          # the network did change
    assert (a1 - a0).abs().max().item() <= 1e-6         # ... and the folded session still holds the old filters
    folded.refresh()
    a2, _ = both()
    assert (a2 - b1).abs().max().item() <= 2e-5


@pytest.mark.parametrize("variant", ["vec", "img"])
def test_batchnorm_apply_inside_the_winograd_input_transform_changes_no_bit(variant, monkeypatch):
    """MMFN_LAZY_BN (default on): the BatchNorm apply (+ skip + ReLU) between two convolutions of a BasicBlock chain runs inside
    the consuming convolution's Winograd input transform (engine.PendingBN; the first BatchNorm output of a block is never
    written, the backward recomputes its ReLU sign).  Same arithmetic, expression for expression: loss, every gradient,
    BatchNorm running statistics and the eval output equal the one-launch-per-BatchNorm step BIT FOR BIT."""
    from mmfn_amd import engine as E
    _, net_a, batch, args = _setup(variant, dropout=0.1)
    _, net_b, _, _ = _setup(variant, dropout=0.1)
    dargs = _dev_args(args)
    gt = batch["gt_wp"].to(DEV)
    net_a.train(); net_b.train()
    inp_a, inp_b = net_a._pack(*dargs), net_b._pack(*dargs)
    monkeypatch.setattr(E, "LAZY_BN_APPLY", False)
    for _ in range(2):
        loss_a = net_a.train_step(inp_a, gt)
    net_a.eval()
    with torch.no_grad():
        out_a = net_a(*dargs)
    monkeypatch.setattr(E, "LAZY_BN_APPLY", True)
    for _ in range(2):
        loss_b = net_b.train_step(inp_b, gt)
    net_b.eval()
    with torch.no_grad():
        out_b = net_b(*dargs)
    torch.cuda.synchronize()
    assert loss_a.item() == loss_b.item()
    La, Lb = net_a._layout, net_b._layout
    assert torch.equal(La.grads[:La.tail], Lb.grads[:Lb.tail]) and torch.equal(La.params, Lb.params)
    for (ka, va), (kb, vb) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    assert torch.equal(out_a, out_b)
    # the fused sites really ran fused: a block's first BatchNorm output was never materialised
    eng = net_b._engine_for()
    blk = eng.img.layers[2][1]
    assert blk.c1.saved[2] is None and blk.c2.x_is_standin and blk.c2.saved[2] is not None


@pytest.mark.parametrize("variant", ["vec", "rad"])
def test_layernorm_inside_the_qkv_and_mlp_gemms_tracks_the_separate_launches(variant, monkeypatch):
    """MMFN_LN_FOLD (fp32 path; default: eval-mode forwards only, "1": training steps too): ln1 -> key/query/value and ln2 -> mlp.0 of every transformer block run as one GEMM
    launch each (MMFN_EPI_LN_FOLD), the normalised tensors the weight gradients need are recomputed on the side stream in the
    backward.  Different rounding, same function: train loss, eval waypoints and the transformer outputs agree with the
    LayerNorm-launch path to fp32 accuracy, the weight gradients of the folded Linears to the accuracy the forward allows, and the
    forward really has no LayerNorm launch per block (the tensors are made in the backward)."""
    from mmfn_amd import engine as E
    from oracle import harness
    oracle, net_a, batch, args = _setup(variant)
    monkeypatch.setattr(E, "LN_FOLD", "0")
    net_a._engine_for()
    monkeypatch.setattr(E, "LN_FOLD", "1")
    _, net_b, _, _ = _setup(variant)
    ea, eb = net_a._engine_for(), net_b._engine_for()
    assert ea.ln_fold_table is None and eb.ln_fold_table is not None
    dargs = _dev_args(args)
    gt = batch["gt_wp"].to(DEV)
    net_a.train(); net_b.train()
    inp_a, inp_b = net_a._pack(*dargs), net_b._pack(*dargs)
    _, la = ea.forward(inp_a, True, gt)
    _, lb = eb.forward(inp_b, True, gt)
    assert eb.gpts[0].folded_fwd and not getattr(ea.gpts[0], "folded_fwd", False)
    assert abs(la.item() - lb.item()) <= 2e-6 * max(1.0, abs(la.item())), (la.item(), lb.item())
    for k in ("gpt1", "gpt2", "gpt3", "gpt4"):
        ta, tb = ea.taps[k], eb.taps[k]
        # (gpt4 sits behind four fusion stages and three ResNet stages of train-mode BatchNorms: with the closed-form fill a 1e-6
        # difference in transformer 1 arrives as ~3e-4 - measured; the loss above and the oracle comparisons below are the bar)
        assert (ta - tb).abs().max().item() <= 1e-3 * ta.abs().max().item(), (k, (ta - tb).abs().max().item(), ta.abs().max().item())
    # the saved statistics are the LayerNorm kernel's to rounding (first transformer: the later ones see inputs that already differ
    # by the amplified rounding noted above)
    for ga, gb in zip(ea.gpts[:1], eb.gpts[:1]):
        for ba, bb in zip(ga.blocks, gb.blocks):
            for ln in ("ln1", "ln2"):
                assert (ba[ln].saved[1] - bb[ln].saved[1]).abs().max().item() <= 1e-5
                assert ((ba[ln].saved[2] - bb[ln].saved[2]) / ba[ln].saved[2]).abs().max().item() <= 5e-5
    ea.backward(); eb.backward()
    torch.cuda.synchronize()
    La, Lb = net_a._layout, net_b._layout
    # at batch 2 with the closed-form fill the backward amplifies the forward's rounding difference: the deep
    # stage, which the amplification has not reached, must agree closely; the folded Linears' own gradients likewise
    b0, e0 = La.stage_ranges[0]
    x, y = La.grads[b0:e0].double(), Lb.grads[b0:e0].double()
    assert float((x * y).sum() / (x.norm() * y.norm())) >= 0.99999
    for name in ("encoder.transformer4.blocks.7.attn.key.weight", "encoder.transformer4.blocks.7.mlp.0.weight",
                 "encoder.transformer4.blocks.7.ln1.weight", "encoder.transformer4.blocks.0.ln2.bias"):
        x, y = La.grad_views[name].double().flatten(), Lb.grad_views[name].double().flatten()
        assert float((x * y).sum() / (x.norm() * y.norm())) >= 0.9999, name
    # eval forward (calibrated running statistics), as the agents run it
    harness.calibrate_bn(oracle, args)
    for net in (net_a, net_b):
        net.load_state_dict(oracle.state_dict(), strict=True)
        net.eval()
    with torch.no_grad():
        ref = oracle(*args)
        oa, ob = net_a(*dargs).cpu(), net_b(*dargs).cpu()
    assert (oa - ref).abs().max().item() <= 1e-4 and (ob - ref).abs().max().item() <= 1e-4
