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
    from mmfn_amd.model import MMFN, MMFNImg
    cls = {"vec": MMFN, "img": MMFNImg}[variant]
    kw = dict(embd_pdrop=dropout, attn_pdrop=dropout, resid_pdrop=dropout)
    torch.manual_seed(seed)
    a = cls(GlobalConfig(**kw), DEV)
    b = cls(GlobalConfig(act_dtype="bf16", **kw), DEV)
    b.load_state_dict(a.state_dict())
    inp, gt = bench.synth_inputs(batch, DEV, seed=seed, variant=variant)
    return a.train(), b.train(), inp, gt


def _stage_cosines(La, Lb):
    out = []
    for b, e in La.stage_ranges:
        x, y = La.grads[b:min(e, La.tail)].double(), Lb.grads[b:min(e, Lb.tail)].double()
        out.append(float((x * y).sum() / (x.norm() * y.norm())))
    return out


@pytest.mark.parametrize("variant,batch", [("vec", 32), ("img", 8), ("vec", 2)])
def test_bf16_mode_tracks_the_fp32_path(variant, batch):
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
    assert dt["gpt4.S.gh"] == torch.bfloat16 and dt["img.l3.1.c2.dconv"] == torch.bfloat16
    cos = _stage_cosines(a._layout, b._layout)
    # stage 0 (fusion scale 4 + head: the gradient before it has passed the deep BatchNorm stacks) must be clean; the others are
    # bounded by what torch.autocast(bfloat16) itself reaches on this network (CPU oracle, same init: 0.998 / 0.84 / 0.79 / 0.79
    # at batch 8 - DESIGN.md section 7), minus a margin for the bf16 residual stream autocast keeps in fp32
    assert cos[0] >= 0.97, cos
    assert min(cos[1:]) >= 0.60, cos
    # eval-mode forward (running statistics): waypoints to bf16 accuracy of their scale
    a.eval(), b.eval()
    with torch.no_grad():
        pa, _ = ea.forward(inp, False, None)
        pb, _ = eb.forward(inp, False, None)
    assert float((pa - pb).abs().max()) <= 5e-2 * float(pa.abs().max())   # ~100 bf16 layers deep


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
    from mmfn_amd.model import MMFNRad
    net = MMFNRad(GlobalConfig(act_dtype="bf16"), DEV)
    with pytest.raises(NotImplementedError):
        net._engine_for()
