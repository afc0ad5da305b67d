"""The bf16 training mode (GlobalConfig(act_dtype="bf16"), BASELINE configs[2] arithmetic): bf16 activations / saved tensors /
weight shadows in HBM, fp32 accumulation, statistics, master weights, gradients and optimizer.

What can be asked of it.  The forward is well conditioned: loss and waypoints must agree with the fp32 path on the same weights
to bf16 accuracy (bars below).  The backward of this network at its initialisation is not: two fp32 evaluations (HIP vs the CPU
oracle, or the oracle vs itself in fp64) already differ by percents per tensor, so a per-tensor gradient
bar in the bf16 mode would test chaos, not kernels.  The gradient is therefore judged (a) per backward stage by its cosine to the
fp32 HIP gradient, against the floor PyTorch's own autocast reaches on the CPU oracle for the same network, and (b) by what it
is for: a few optimizer steps from it must reduce the loss like the fp32 steps do.  Everything deterministic is tested exactly:
graph replay == eager steps bit for bit, and the kernels one by one against torch (tests/test_gemm16_gpu.py, test_kernels16)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _pair(batch, variant="vec", dropout=0.0, seed=42):
    import bench
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN, MMFNImg, MMFNRad
    n_views = 2 if variant == "img2v" else 1          # img2v: two camera views = 256 tokens per sample, like the rad variant's
    variant = "img" if variant == "img2v" else variant
    cls = {"vec": MMFN, "img": MMFNImg, "rad": MMFNRad}[variant]
    kw = dict(embd_pdrop=dropout, attn_pdrop=dropout, resid_pdrop=dropout, n_views=n_views)
    torch.manual_seed(seed)
    a = cls(GlobalConfig(**kw), DEV)
    b = cls(GlobalConfig(act_dtype="bf16", **kw), DEV)
    b.load_state_dict(a.state_dict())
    inp, gt = bench.synth_inputs(batch, DEV, seed=seed, variant=variant, n_views=n_views)
    return a.train(), b.train(), inp, gt


def _stage_cosines(La, Lb):
    out = []
    for b, e in La.stage_ranges:
        x, y = La.grads[b:min(e, La.tail)].double(), Lb.grads[b:min(e, Lb.tail)].double()
        out.append(float((x * y).sum() / (x.norm() * y.norm())))
    return out


@pytest.mark.parametrize("variant,batch", [("vec", 32), ("img", 8), ("vec", 2), ("rad", 8), ("img2v", 4)])
def test_bf16_mode_tracks_the_fp32_path(variant, batch):
    """(rad: the radar GAT stays an fp32 island, its feature joins the deepest fusion as a bf16 activation; its transformer and
    img2v's four run the 256-token instantiation of attention16.hip.)"""
    a, b, inp, gt = _pair(batch, variant)
    ea, eb = a._engine_for(), b._engine_for()
    _, la = ea.forward(inp, True, gt)
    ea.backward()
    _, lb = eb.forward(inp, True, gt)
    eb.backward()
    torch.cuda.synchronize()
    la, lb = float(la.item()), float(lb.item())
    assert abs(la - lb) <= 2e-3 * abs(la), (la, lb)                 # the bar the round-2 review set: loss within 2e-3 relative
    # activations really are bf16 in HBM, the fp32 islands really are fp32
    bufs = eb._bufs_for(batch)._bufs
    dt = {k[0]: v.dtype for k, v in bufs.items()}
    assert dt["img.l2.0.c1.out"] == torch.bfloat16 and dt["img.l2.0.c1.conv"] == torch.bfloat16 and dt["gpt4.b0.qkv"] == torch.bfloat16
    assert dt["img.stem.col"] == torch.bfloat16 and dt["img.stem.conv"] == torch.bfloat16 and dt["img.stem.out"] == torch.bfloat16
    assert dt["in.img"] == torch.float32 and dt["fused"] == torch.float32
    if variant == "vec":
        assert dt["vec.gen.pre"] == torch.float32 and dt["vec.out"] == torch.bfloat16
    if variant == "rad":
        assert dt["rad.out"] == torch.float32 and dt["rad.out16"] == torch.bfloat16 and eb.gpts[3].T == 256 and eb.gpts[2].T == 192
    if variant == "img2v":
        assert all(g.T == 256 for g in eb.gpts)
    assert dt["gpt4.S.gh"] == torch.bfloat16 and dt["img.l3.1.c2.dconv"] == torch.bfloat16
    # ... and the transformers' residual stream and its gradient stay fp32, like torch.autocast's (x + Linear(LN(x)))
    assert dt["gpt4.b0.x1"] == torch.float32 and dt["gpt4.x0"] == torch.float32 and dt["gpt4.S.g"] == torch.float32
    assert dt["gpt4.S.a"] == torch.bfloat16 and dt["gpt4.S.gdrop"] == torch.bfloat16
    cos = _stage_cosines(a._layout, b._layout)
    print("\n[%s B=%d] bf16 mode vs fp32 path: loss %.6f / %.6f, per-stage gradient cosine %s" % (
        variant, batch, lb, la, " ".join("%.4f" % c for c in cos)))
    # sanity floors only: the yardstick test is test_bf16_mode_gradient_direction_against_torch_autocast below (measured at
    # vec B=32: 0.994 / 0.838 / 0.797 / 0.789; round 3, with a bf16 residual stream in the transformers: 0.989 / 0.763 / 0.715 / 0.718)
    assert cos[0] >= (0.985 if batch >= 8 else 0.97), cos
    assert min(cos[1:]) >= (0.70 if batch >= 8 else 0.55), cos
    # eval-mode forward: waypoints to bf16 accuracy of their scale.  One train-mode forward with BatchNorm momentum 1.0 first
    # (running statistics := batch statistics, as oracle/make_golden.py does): with the freshly initialised running statistics
    # (0, 1) the eval network runs 85 BatchNorms far off their operating point and any rounding difference is amplified
    import torch.nn as nn
    for net, eng in ((a, ea), (b, eb)):
        bns = [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)]
        for m in bns:
            m.momentum = 1.0
        eng.forward(inp, True, gt)
        for m in bns:
            m.momentum = 0.1
    a.eval(), b.eval()
    with torch.no_grad():
        pa, _ = ea.forward(inp, False, None)
        pb, _ = eb.forward(inp, False, None)
    err, scale = float((pa - pb).abs().max()), float(pa.abs().max())
    assert err <= 5e-2 * scale, (err, scale)   # ~100 bf16 layers deep


@pytest.mark.parametrize("batch", [8, 32])
def test_bf16_mode_gradient_direction_against_torch_autocast(batch):
    """The yardstick the round-3 review asked for.  What does bf16 arithmetic do to THIS network's gradient?  PyTorch's own
    answer: the CPU oracle under torch.autocast(bfloat16) against itself in fp32 - same reference-style initialisation
    (seed 42), same batch, per backward stage.  The HIP bf16 mode against the HIP fp32 path must reach that cosine - 0.03 in
    every stage (measured: it is ABOVE autocast in every stage - batch 8: 0.9985 / 0.856 / 0.816 / 0.804 against 0.9982 / 0.839 /
    0.795 / 0.783 - since the transformers' residual stream stays fp32 like autocast's; with the bf16 stream of round 3 the mode
    sat 0.035-0.08 BELOW it, and the CPU experiment tools/experiments/bf16_where.py reproduces that drop by rounding the stream
    alone).  And the loss of the bf16 step against the ORACLE (fp32, CPU): within 2e-3 relative."""
    import bench
    from mmfn_amd.params import FlatLayout
    from oracle import harness
    torch.set_num_threads(bench.usable_cores())
    a, b, inp, gt = _pair(batch)
    ea, eb = a._engine_for(), b._engine_for()
    _, la = ea.forward(inp, True, gt)
    ea.backward()
    _, lb = eb.forward(inp, True, gt)
    eb.backward()
    torch.cuda.synchronize()
    cos_hip = _stage_cosines(a._layout, b._layout)
    oracle = harness.build_oracle("vec", dropout=0.0)
    oracle.load_state_dict({k: v.detach().cpu() for k, v in a.state_dict().items()}, strict=True)   # the weights before the step
    cpu = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    args = harness.forward_args(bench.oracle_batch_from_inputs(cpu, "vec"), "vec")

    def oracle_grads(autocast):
        oracle.train()
        for p in oracle.parameters():
            p.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            pred = oracle(*args)
        loss = harness.l1_waypoint_loss(pred.float(), gt.cpu())
        loss.backward()
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in oracle.named_parameters() if p.grad is not None}

    # (BatchNorm running statistics move in these train-mode forwards; the gradients do not depend on them)
    lo32, go32 = oracle_grads(False)
    lo16, go16 = oracle_grads(True)
    cos_ref = []
    for st in range(4):
        names = [k for k in go32 if FlatLayout.stage_of(k) == st]
        x = torch.cat([go32[k].flatten().double() for k in names])
        y = torch.cat([go16[k].flatten().double() for k in names])
        cos_ref.append(float(torch.dot(x, y) / (x.norm() * y.norm())))
    print("\n[vec B=%d] per-stage gradient cosine to the fp32 gradient: HIP bf16 mode %s | torch.autocast(bfloat16) on the CPU oracle %s"
          % (batch, " ".join("%.4f" % c for c in cos_hip), " ".join("%.4f" % c for c in cos_ref)))
    print("    loss: HIP bf16 %.6f  HIP fp32 %.6f  oracle fp32 %.6f  oracle autocast %.6f" % (float(lb), float(la), lo32, lo16))
    for st in range(4):
        assert cos_hip[st] >= cos_ref[st] - 0.03, (st, cos_hip, cos_ref)
    assert abs(float(lb) - lo32) <= 2e-3 * abs(lo32), (float(lb), lo32)
    assert abs(float(la) - lo32) <= 1e-4


def test_bf16_steps_reduce_the_loss_like_fp32_steps():
    a, b, inp, gt = _pair(16)
    la = [float(a.train_step(inp, gt, lr=1e-4).item()) for _ in range(12)]
    lb = [float(b.train_step(inp, gt, lr=1e-4).item()) for _ in range(12)]
    assert la[-1] < 0.97 * la[0] and lb[-1] < 0.97 * lb[0]
    # same trajectory within a few percent of the total descent at every step
    drop = la[0] - la[-1]
    assert max(abs(x - y) for x, y in zip(la, lb)) <= 0.25 * drop, (la, lb)


def test_bf16_graph_replay_equals_eager_steps():
    from mmfn_amd.parallel import GraphedStep
    a, b, inp, gt = _pair(4, dropout=0.1)
    del a
    import copy
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    torch.manual_seed(1)
    c = MMFN(GlobalConfig(act_dtype="bf16"), DEV).train()
    c.load_state_dict(b.state_dict())
    c._engine_for().rng_state.copy_(b._engine_for().rng_state)
    for _ in range(3):
        lb = b.train_step(inp, gt)
    step = GraphedStep(c._engine_for(), None, inp, gt, warm=1)
    for _ in range(2):
        lc = step()
    torch.cuda.synchronize()
    assert float(lb.item()) == float(lc.item())
    assert torch.equal(b._layout.params, c._layout.params)


def test_bf16_mode_rejects_what_it_does_not_cover():
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFNImg
    net = MMFNImg(GlobalConfig(act_dtype="bf16", n_views=3), DEV)   # 320 tokens: K and V of a 128-wide head no longer fit LDS
    with pytest.raises(NotImplementedError, match="256 tokens"):
        net._engine_for()


@pytest.mark.parametrize("variant", ["vec", "img"])
def test_bf16_closed_loop_session_over_folded_batchnorms(variant):
    """DrivingSession in the bf16 mode: the eval-mode BatchNorms folded into bf16 shadows of the filters (conv + shift + skip + ReLU in
    one launch of the implicit GEMM), the tick one hipGraph.  Against the unfolded eager bf16 session (the folded filter w * s is
    rounded to bf16 once instead of scaling fp32 accumulators: a bf16-sized difference), against the fp32 session (the mode's eval
    bar: 5e-2 of the waypoint scale), and refresh() follows changed weights without a new capture."""
    import numpy as np
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.inference import DrivingSession
    from mmfn_amd.model import MMFN, MMFNImg
    cls = {"vec": MMFN, "img": MMFNImg}[variant]
    torch.manual_seed(7)
    net32 = cls(GlobalConfig(), DEV)
    with torch.no_grad():   # running statistics away from their initial (0, 1)
        for name, b in net32.named_buffers():
            if name.endswith("running_var"):
                b.uniform_(0.5, 1.5)
            elif name.endswith("running_mean"):
                b.uniform_(-0.2, 0.2)
    net16 = cls(GlobalConfig(act_dtype="bf16"), DEV)
    net16.load_state_dict(net32.state_dict())
    ref = DrivingSession(net32.eval(), max_points=1 << 14, max_lanes=16)
    folded = DrivingSession(net16.eval(), max_points=1 << 14, max_lanes=16)
    plain = DrivingSession(net16, max_points=1 << 14, max_lanes=16, fold_batchnorm=False, use_graph=False)
    assert folded.fold and folded.eng.act_dtype == torch.bfloat16
    rng = np.random.RandomState(1)
    rgb = rng.randint(0, 256, (300, 400, 3)).astype(np.uint8)
    pts = np.stack([rng.uniform(-20, 20, 5000), rng.uniform(-12, 28, 5000), rng.uniform(-3, 1, 5000), rng.uniform(0, 1, 5000)], 1).astype(np.float32)
    lanes = rng.randn(6, 10, 5).astype(np.float32)
    bev = rng.randint(0, 256, (256, 256, 3)).astype(np.uint8)

    def run(sess):
        kw = dict(merge_previous_sweep=False)
        if variant == "img":
            return sess.predict(rgb, pts, None, (2.0, 15.0), 3.0, map_image=bev, **kw).float().clone()
        return sess.predict(rgb, pts, lanes, (2.0, 15.0), 3.0, **kw).float().clone()

    r, a, b = run(ref), run(folded), run(plain)
    scale = float(r.abs().max()) + 1e-6
    dab, dar, dbr = float((a - b).abs().max()), float((a - r).abs().max()), float((b - r).abs().max())
    print("\n[bf16 session %s] scale %.4g: folded-unfolded %.3g, folded-fp32 %.3g, unfolded-fp32 %.3g" % (variant, scale, dab, dar, dbr))
    assert dar <= 5e-2 * scale and dab <= 5e-2 * scale, (dab, dar, dbr, scale)       # the mode's eval bar
    assert dar <= 2.5 * dbr + 1e-2 * scale, (dab, dar, dbr, scale)                   # folding costs no more than the mode itself
    assert torch.equal(a, run(folded))                                   # the replayed tick is deterministic
    with torch.no_grad():
        for name, p in net16.named_parameters():
            if "image_encoder" in name and name.endswith("conv1.weight"):
                p.mul_(1.05)
    net16.weights_changed()
    b1 = run(plain)
    assert float((b1 - b).abs().max()) > 1e-3 * scale                   # the network did change
    a1 = run(folded)                                                     # (predict() re-folds when the module's weight version moved)
    scale1 = float(b1.abs().max()) + 1e-6
    print("[bf16 session %s] after the weight change: scale %.4g, folded-unfolded %.3g, moved by %.3g" % (variant, scale1, float((a1 - b1).abs().max()),
                                                                                                     float((b1 - b).abs().max())))
    assert float((a1 - b1).abs().max()) <= 5e-2 * scale1
