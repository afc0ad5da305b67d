"""Oracle parity AT THE SIZES bench.py RUNS (BASELINE.json configs[1], [3], [4]) - not only at batch 2.

  configs[1]  vec, batch 32, 16384-point LiDAR, 64 lanes       raw u8 frames + XYZI points through ingest / splat
  configs[4]  rad, batch 16, 65536-point LiDAR, 64 lanes + radar
  configs[3]  ResNet-34 camera branch alone, batch 128          against a torch fp32 ResNet-34 (oracle/model.py)

The oracle (plain PyTorch fp32 on the host cores) needs ~5-20 s per step at these sizes.  north_star bar: waypoint L1
loss within 1e-4 of the reference CPU path on identical batches; waypoints within 1e-4.  Gradients: two fp32 evaluations
of this graph (85 train-mode BatchNorms in the backward) differ by ~1e-2 relative per tensor even at batch 32, so - as in
tests/test_e2e_gpu.py - the HIP gradient is judged against an fp64 oracle with the fp32 oracle's own error on the same
tensor as the yardstick.  This also exercises every (tile, split-K) entry of the tuning table that only the large shapes
select."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _threads():
    import bench
    torch.set_num_threads(bench.usable_cores())


def _build(variant, B, n_lidar, lanes=64):
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd import model as M
    from oracle import fixtures, harness
    _threads()
    oracle = harness.build_oracle(variant, dropout=0.0)
    net = {"vec": M.MMFN, "img": M.MMFNImg, "rad": M.MMFNRad}[variant](
        GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0), DEV)
    net.load_state_dict(oracle.state_dict(), strict=True)
    batch = fixtures.synthetic_batch(B, variant, seed=42, n_lidar=n_lidar, lanes=lanes)
    args = harness.forward_args(batch, variant)   # the reference's CPU preprocessing (crop, np.histogramdd)
    return oracle, net, batch, args


def _raw_inputs(batch, variant):
    """Engine inputs exactly as bench.py feeds them: raw u8 camera frames and XYZI points resident on the device."""
    to = lambda t: t.to(DEV).contiguous()
    inp = {"rgb_u8": to(batch["rgb_u8"]), "lidar_pts": to(batch["lidar_pts"]), "lane": to(batch["lane"]),
           "lane_num": to(batch["lane_num"].to(torch.int32)), "target_point": to(batch["target_point"]),
           "velocity": to(batch["velocity"])}
    if variant == "rad":
        inp["radar"], inp["radar_adj"] = to(batch["radar"]), to(batch["radar_adj"])
    return inp


def _to64(a):
    if torch.is_tensor(a):
        return a.double() if a.is_floating_point() else a
    if isinstance(a, (list, tuple)):
        return type(a)(_to64(x) for x in a)
    return a


def _grad_report(hip_grads, g32, g64):
    """Per-parameter error of the HIP gradient and of the fp32 oracle's gradient, both against an fp64 evaluation of the
    same graph: (|hip - f64| / |f64|, |cpu32 - f64| / |f64|, cosine(hip, f64), name)."""
    rows = []
    gmax = max(t.norm().item() for t in g64.values() if t is not None)
    for name, t in g64.items():
        a = hip_grads[name]
        if t is None:
            assert a is None, name
            continue
        assert a is not None, name
        a = a.detach().cpu().double().flatten()
        b = t.flatten()
        n = b.norm().item()
        if n <= 1e-7 * gmax:   # exactly-zero / noise-level tensors (pos_emb.0.weight: zero input)
            assert a.norm().item() <= 1e-5 * gmax, name
            continue
        c = g32[name].double().flatten()
        rows.append(((a - b).norm().item() / n, (c - b).norm().item() / n, float(torch.dot(a, b) / (a.norm() * b.norm())), name))
    return rows


def _stage_cosines_vs_fp64(hip_grads, g32, g64):
    """Per BACKWARD STAGE (params.FlatLayout.stage_of: fusion scale 4 first), over all tensors of the stage together: cosine of
    the HIP gradient and of the fp32 oracle's gradient to the fp64 gradient.  A systematic (non-noise) kernel error - a wrong
    scale, a dropped term, a transposed filter - moves the direction of a whole stage; rounding noise that the per-tensor
    error-ratio test has to tolerate does not (round-2 review, item 8)."""
    from mmfn_amd.params import FlatLayout
    acc = {}
    for name, t in g64.items():
        if t is None:
            continue
        st = FlatLayout.stage_of(name)
        a, b, c = hip_grads[name].detach().cpu().double().flatten(), t.flatten(), g32[name].double().flatten()
        d = acc.setdefault(st, [0.0] * 5)
        d[0] += float(torch.dot(a, b)); d[1] += float(torch.dot(a, a)); d[2] += float(torch.dot(b, b))
        d[3] += float(torch.dot(c, b)); d[4] += float(torch.dot(c, c))
    return {st: (d[0] / (d[1] * d[2]) ** 0.5, d[3] / (d[4] * d[2]) ** 0.5) for st, d in sorted(acc.items())}


def _judge_stage_cosines(cos, tag):
    print("[%s] per-stage gradient cosine to fp64 (HIP, CPU fp32 oracle): %s"
          % (tag, "  ".join("stage %d: %.6f, %.6f" % (st, h, c) for st, (h, c) in cos.items())))
    for st, (h, c) in cos.items():
        # 0.9999 wherever the fp32 oracle itself gets there; otherwise the HIP gradient may be at most 3x further (in angle^2)
        # from the fp64 direction than the fp32 oracle is
        bar = min(0.9999, 1.0 - 3.0 * (1.0 - c) - 1e-6)
        assert h >= bar, "stage %d: cosine(HIP, fp64) %.6f below %.6f (fp32 oracle %.6f)" % (st, h, bar, c)


def _judge_gradients(rows, tag):
    """Both the HIP path and the CPU oracle are fp32 evaluations of a graph whose backward amplifies rounding (85 train-mode
    BatchNorms): the yardstick for the HIP error is the fp32 oracle's OWN error against fp64 on the same tensor."""
    rel_hip = sorted(r[0] for r in rows)
    rel_cpu = sorted(r[1] for r in rows)
    ratios = sorted(r[0] / max(r[1], 1e-9) for r in rows)
    worst = max(rows)
    print("[%s] gradient error vs fp64: HIP median %.2e p95 %.2e max %.2e (%s) | CPU fp32 oracle median %.2e p95 %.2e max %.2e | "
          "ratio HIP/CPU median %.2f p95 %.2f | min cosine %.6f"
          % (tag, rel_hip[len(rows) // 2], rel_hip[int(len(rows) * 0.95)], worst[0], worst[3], rel_cpu[len(rows) // 2],
             rel_cpu[int(len(rows) * 0.95)], rel_cpu[-1], ratios[len(rows) // 2], ratios[int(len(rows) * 0.95)], min(r[2] for r in rows)))
    assert ratios[len(rows) // 2] <= 2.5, "median HIP / CPU-fp32 gradient error ratio %g" % ratios[len(rows) // 2]
    assert ratios[int(len(rows) * 0.95)] <= 8.0, "p95 HIP / CPU-fp32 gradient error ratio %g" % ratios[int(len(rows) * 0.95)]
    # per tensor: 12x the fp32 oracle's error on that tensor, or on a typical tensor where the oracle happened to land
    # unusually close to fp64 (the error ratio is long-tailed in both directions).  No cosine bound: with this closed-form
    # weight fill some tensors' fp32 gradients - the oracle's included - point opposite to the fp64 ones (cosine -0.99999).
    med_cpu = rel_cpu[len(rows) // 2]
    bad = [r for r in rows if r[0] > 12.0 * max(r[1], med_cpu) + 2e-4]
    assert not bad, "per-tensor bound (|hip-f64|, |cpu32-f64|, cos, name): %s" % bad[:6]


def _check_train_step(variant, B, n_lidar):
    import copy
    from oracle import harness
    oracle, net, batch, args = _build(variant, B, n_lidar)
    o64 = copy.deepcopy(oracle).double()
    _, loss64, g64 = harness.train_step(o64, _to64(args), batch["gt_wp"].double())
    del o64
    pred_ref, loss_ref, grads_ref = harness.train_step(oracle, args, batch["gt_wp"])
    net.train()
    eng = net._engine_for()
    inp = _raw_inputs(batch, variant)
    gt = batch["gt_wp"].to(DEV)
    pred, loss = eng.forward(inp, True, gt)
    eng.backward()
    net._layout.attach_grads()
    torch.cuda.synchronize()
    wp_err = (pred.cpu() - pred_ref).abs().max().item()
    loss_err = abs(loss.item() - loss_ref.item())
    print("\n[%s B=%d N=%d] loss hip %.7f oracle fp32 %.7f fp64 %.7f |hip-fp32| %.2e; waypoint max err %.2e"
          % (variant, B, n_lidar, loss.item(), loss_ref.item(), loss64.item(), loss_err, wp_err))
    assert loss_err <= 1e-4, (loss.item(), loss_ref.item())          # north_star: within 1e-4 fp32
    assert abs(loss.item() - loss64.item()) <= 1e-4
    assert wp_err <= 1e-4 * max(1.0, pred_ref.abs().max().item()), wp_err
    hip_grads = {n: p.grad for n, p in net.named_parameters()}
    _judge_gradients(_grad_report(hip_grads, grads_ref, g64), "%s B=%d" % (variant, B))
    _judge_stage_cosines(_stage_cosines_vs_fp64(hip_grads, grads_ref, g64), "%s B=%d" % (variant, B))
    # one fused AdamW step from these gradients == torch.optim.AdamW on the oracle, on the elements whose gradient sign both
    # fp32 evaluations determine
    init = {k: v.detach().clone() for k, v in net.named_parameters()}
    hip_g = {k: v.grad.detach().cpu().double() for k, v in net.named_parameters() if v.grad is not None}
    eng.optimizer_step(lr=1e-4)
    torch.cuda.synchronize()
    ref_sd = oracle.state_dict()
    checked = total = 0
    gmax = max(t.norm().item() for t in g64.values() if t is not None)
    for name, p in net.named_parameters():
        t = g64[name]
        if t is None:
            continue
        sure = t.abs() > 10.0 * torch.maximum((grads_ref[name].double() - t).abs(), (hip_g[name] - t).abs()) + 1e-7 * gmax
        upd = (p.detach().cpu().double() - init[name].cpu().double())
        upd_ref = ref_sd[name].double() - init[name].cpu().double()
        total += t.numel()
        checked += int(sure.sum())
        if sure.any():
            d = (upd - upd_ref)[sure].abs().max().item()
            assert d <= 2e-6, (name, d)   # lr 1e-4: a wrong sign is 2e-4, a missing update 1e-4
    assert checked > 0.05 * total, (checked, total)   # vec B=32 with this weight fill: ~16 % of the elements qualify
    # BatchNorm running statistics after the step (momentum 0.1, unbiased variance)
    got_sd = net.state_dict()
    for k, v in ref_sd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            d = (got_sd[k].cpu() - v).abs().max().item()
            assert d <= 1e-3 * max(1.0, v.abs().max().item()), (k, d)   # batch variances after ~30 fp32 layers
        elif k.endswith("num_batches_tracked"):
            assert int(got_sd[k].item()) == int(v.item()), k


def test_vec_batch32_16384pts_matches_oracle():
    """BASELINE configs[1] - the headline bench workload."""
    _check_train_step("vec", 32, 16384)


def test_vec_batch32_gradient_direction_at_the_benched_initialisation(monkeypatch):
    """The gradient check that can FAIL (round-3 review, item 2).  At the closed-form weight fill of the golden fixtures the
    backward is so ill-conditioned that the fp32 oracle itself is at cosine 0.57 to fp64 in the shallow stage: no bar there tells
    a wrong kernel from noise.  At the initialisation bench.py trains from - torch.manual_seed(42) + the model class's own init
    (run_steps/utils.py:77-84, model_vec.py:164-177) - the same graph is well conditioned (oracle 0.99998): here the HIP gradient
    must stay within 4x the oracle's own angle^2 to the fp64 gradient (phase2_train_net.py:104-108) per backward stage AND per
    (stage, trunk / transformer / VectorNet) group, and a deliberately broken data gradient of ONE layer1 convolution (two
    frequency slices of its transformed filter swapped between forward and backward) must turn the test red."""
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd import model as M
    from oracle import fixtures, gradcheck, harness
    _threads()
    torch.manual_seed(42)
    net = M.MMFN(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0), DEV)
    oracle = harness.build_oracle("vec", dropout=0.0)
    oracle.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
    batch = fixtures.synthetic_batch(32, "vec", seed=42, n_lidar=16384, lanes=64)
    args = harness.forward_args(batch, "vec")
    loss32, g32, loss64, g64, _ = gradcheck.oracle_gradients(oracle, args, batch["gt_wp"])
    net.train()
    eng = net._engine_for()
    inp = _raw_inputs(batch, "vec")
    gt = batch["gt_wp"].to(DEV)

    def hip_gradients(fault=None):
        _, loss = eng.forward(inp, True, gt)
        if fault is not None:
            fault()
        eng.backward()
        net._layout.attach_grads()
        torch.cuda.synchronize()
        return loss.item(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}

    def judge(hip, tag):
        bad = []
        for what, group in (("stage", gradcheck.stage_of), ("group", gradcheck.group_of)):
            cos = gradcheck.stage_cosines({k: hip.get(k) for k in g64}, g32, g64, group=group)
            print("[%s] cosine to the fp64 gradient per %s (HIP / CPU fp32 oracle / bar): %s" % (
                tag, what, "  ".join("%s: %.6f / %.6f / %.6f" % (k, h, c, gradcheck.stage_bar(c)) for k, (h, c) in cos.items())))
            bad += [(k, h, c) for k, (h, c) in cos.items() if h < gradcheck.stage_bar(c)]
        return bad

    loss, hip = hip_gradients()
    print("\n[vec B=32, benched initialisation] loss HIP %.7f  CPU fp32 %.7f  fp64 %.7f" % (loss, float(loss32), float(loss64)))
    assert abs(loss - float(loss32)) <= 1e-4 and abs(loss - float(loss64)) <= 1e-4
    bad = judge(hip, "as built")
    assert not bad, "gradient direction outside the bar (key, HIP, oracle): %s" % bad

    # ---- the same check must fail for a wrong backward: break ONE convolution's data gradient
    conv = eng.img.layers[1][1].c1          # camera trunk, layer1, second block, first convolution
    assert conv.saved_u is not None

    def swap_two_frequencies():
        u = conv.saved_u.view(36, -1)
        tmp = u[7].clone()
        u[7].copy_(u[8])
        u[8].copy_(tmp)

    _, hip_broken = hip_gradients(swap_two_frequencies)
    bad = judge(hip_broken, "layer1 data gradient broken")
    assert any(k == (3, "img") for k, _, _ in bad), "a broken layer1 data gradient went unnoticed: %s" % bad
    # (the forward re-derives the transformed filters from the weights, so the next step is clean again)
    _, hip_again = hip_gradients()
    assert all(torch.equal(hip[k], hip_again[k]) for k in hip)


def test_rad_batch16_65536pts_matches_oracle():
    """BASELINE configs[4]: four-modality model, 65536-point sweeps (the splat at its stress size)."""
    _check_train_step("rad", 16, 65536)


def test_splat_65536_points_bit_exact():
    """lidar_to_histogram_features (dataloader.py:271-293) at 65536 points x batch 16: integer counts, bit-exact."""
    from mmfn_amd import ops
    from oracle import fixtures, preprocess
    batch = fixtures.synthetic_batch(16, "vec", seed=7, n_lidar=65536, lanes=4)
    pts = batch["lidar_pts"]
    ref = np.stack([preprocess.lidar_histogram(p[:, :3].numpy().astype(np.float64)) for p in pts])  # [B,2,256,256]
    out = ops.lidar_splat(pts.to(DEV).contiguous(), torch.empty(16, 256, 256, 2, device=DEV))
    got = out.permute(0, 3, 1, 2).cpu().numpy()
    assert np.array_equal(got, ref)
    assert got.max() == 1.0 and (got > 0).mean() > 0.2   # the clip at 5 points per cell is exercised


def test_image_branch_batch128_matches_torch_resnet34():
    """BASELINE configs[3]: the ResNet-34 camera branch alone at batch 128 (bench.py --workload image-only) against the
    oracle's torch fp32 ResNet-34: pooled features, and every conv / BN parameter gradient of the branch."""
    import bench
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    from oracle import fixtures, harness, preprocess
    from oracle.model import normalize_imagenet
    _threads()
    B = 128
    oracle = harness.build_oracle("vec", dropout=0.0)
    net = MMFN(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0), DEV)
    net.load_state_dict(oracle.state_dict(), strict=True)
    net.train()
    g = torch.Generator().manual_seed(3)
    rgb = torch.randint(0, 256, (B, 300, 400, 3), generator=g, dtype=torch.uint8)
    step = bench.ImageBranchOnly(net._engine_for(), rgb.to(DEV))
    pooled = step()
    net._layout.attach_grads()
    torch.cuda.synchronize()
    # torch reference: same crop, normalisation, trunk, global average pool, loss = mean(pooled) - in fp32 (the oracle) and
    # in fp64 (the yardstick for both fp32 evaluations)
    import copy
    x = torch.from_numpy(np.stack([preprocess.crop_chw(im) for im in rgb.numpy()]).copy()).float()
    x = normalize_imagenet(x)

    def run(trunk, x):
        trunk.train()
        for p in trunk.parameters():
            p.grad = None
        f = trunk.stem(x)
        for li in range(1, 5):
            f = getattr(trunk, "layer%d" % li)(f)
        pooled = f.mean((2, 3))
        pooled.mean().backward()
        return pooled.detach(), {"encoder.image_encoder.features." + k: p.grad for k, p in trunk.named_parameters()}

    trunk = oracle.encoder.image_encoder.features
    t64 = copy.deepcopy(trunk).double()
    pooled64, g64 = run(t64, x.double())
    ref_pooled, g32 = run(trunk, x)
    err = (pooled.cpu().double() - pooled64).abs().max().item()
    err_cpu = (ref_pooled.double() - pooled64).abs().max().item()
    print("\n[image-only B=128] pooled max err vs fp64: HIP %.2e, CPU fp32 oracle %.2e" % (err, err_cpu))
    # 36 convolutions + train-mode BatchNorms deep: the fp32 oracle itself is this far from fp64
    assert err <= max(4.0 * err_cpu, 1e-4 * max(1.0, pooled64.abs().max().item())), (err, err_cpu)
    params = dict(net.named_parameters())
    rows = _grad_report({k: params[k].grad for k in g64}, g32, g64)
    assert len(rows) >= 100
    _judge_gradients(rows, "image-only B=128")


def _stage_cosines(g_a, g_b, layout):
    out = []
    for b, e in layout.stage_ranges:
        e = min(e, layout.tail)
        a, c = g_a[b:e].double(), g_b[b:e].double()
        out.append(float(torch.dot(a, c) / (a.norm() * c.norm())))
    return out


def test_bf16_mode_tracks_fp32_at_least_as_well_as_torch_autocast():
    """BASELINE configs[2] arithmetic (GlobalConfig(gemm_dtype="bf16"): bf16 MFMA operands for every Linear and - as direct
    implicit GEMMs - every convolution except the two 7x7 stems and the stride-2 data gradients; fp32 accumulation,
    activations, normalisation statistics, master weights and optimizer).

    Tolerance statement.  bf16 carries 8 significand bits, and this network's backward amplifies rounding (tests above: even
    fp32 vs fp64 differs by percents per tensor), so 'parity' for the bf16 mode is defined against what PyTorch's OWN mixed
    precision does to the same network: the CPU oracle under torch.autocast(bfloat16) against itself in fp32, same
    reference-style initialisation (seed 42, run_steps/utils.py:77-84), same batch.  Required:
      * loss within 2e-3 relative of the fp32 HIP path (measured ~2e-6 at batch 32),
      * per backward stage, cosine(g_bf16, g_fp32) of the HIP path >= the oracle's autocast cosine - 0.03
        (measured: HIP 0.995 / 0.88 / 0.85 / 0.85, autocast 0.998 / 0.84 / 0.79 / 0.79),
      * at the bench size (batch 32): loss within 2e-3, deepest-stage cosine >= 0.98, every stage >= 0.75."""
    import bench
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    from mmfn_amd.params import FlatLayout
    from oracle import harness
    _threads()
    dev = torch.device(DEV)

    def hip_grads(dtype, B):
        torch.manual_seed(42)
        net = MMFN(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, gemm_dtype=dtype), dev)
        inp, gt = bench.synth_inputs(B, dev, seed=42)
        eng = net._engine_for()
        net.train()
        _, loss = eng.forward(inp, True, gt)
        eng.backward()
        torch.cuda.synchronize()
        L = net._layout
        return float(loss.item()), L.grads[:L.tail].clone(), L, net

    # ---- batch 8: HIP bf16 vs HIP fp32, against the oracle's autocast-vs-fp32 yardstick on the same batch and weights
    l32, g32, L, net32 = hip_grads("f32", 8)
    l16, g16, _, _ = hip_grads("bf16", 8)
    cos_hip = _stage_cosines(g32, g16, L)
    oracle = harness.build_oracle("vec", dropout=0.0)
    torch.manual_seed(42)
    oracle.load_state_dict({k: v.detach().cpu() for k, v in MMFN(GlobalConfig(), "cpu").state_dict().items()}, strict=True)
    inp, gt = bench.synth_inputs(8, torch.device("cpu"), seed=42)
    args = harness.forward_args(bench.oracle_batch_from_inputs(inp, "vec"), "vec")

    def oracle_grads(autocast):
        oracle.train()
        for p in oracle.parameters():
            p.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            pred = oracle(*args)
        loss = harness.l1_waypoint_loss(pred.float(), gt)
        loss.backward()
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in oracle.named_parameters() if p.grad is not None}

    lo32, go32 = oracle_grads(False)
    lo16, go16 = oracle_grads(True)
    cos_ref = []
    for st in range(4):
        names = [k for k in go32 if FlatLayout.stage_of(k) == st]
        a = torch.cat([go32[k].flatten().double() for k in names])
        c = torch.cat([go16[k].flatten().double() for k in names])
        cos_ref.append(float(torch.dot(a, c) / (a.norm() * c.norm())))
    print("\n[bf16 B=8] loss hip fp32 %.6f bf16 %.6f | oracle fp32 %.6f autocast %.6f" % (l32, l16, lo32, lo16))
    print("[bf16 B=8] per-stage cosine(g_bf16, g_fp32): HIP %s | torch autocast (CPU oracle) %s"
          % (["%.4f" % c for c in cos_hip], ["%.4f" % c for c in cos_ref]))
    assert abs(l32 - lo32) <= 1e-4                      # the fp32 paths agree (same weights, same batch)
    assert abs(l16 - l32) <= 2e-3 * abs(l32)
    for st in range(4):
        assert cos_hip[st] >= cos_ref[st] - 0.03, (st, cos_hip, cos_ref)
    # ---- batch 32 (the bench size)
    l32, g32, L, _ = hip_grads("f32", 32)
    l16, g16, _, _ = hip_grads("bf16", 32)
    cos = _stage_cosines(g32, g16, L)
    print("[bf16 B=32] loss fp32 %.6f bf16 %.6f (rel %.1e); per-stage cosine %s" % (l32, l16, abs(l16 - l32) / abs(l32), ["%.4f" % c for c in cos]))
    assert abs(l16 - l32) <= 2e-3 * abs(l32) and torch.isfinite(g16).all()
    assert cos[0] >= 0.98 and min(cos) >= 0.75, cos
