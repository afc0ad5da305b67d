"""BASELINE.json configs[1] size (batch 32, 16384-point LiDAR, 64 lanes): properties that need no CPU oracle run.

The oracle takes minutes per step at this size, so these tests pin the full-size path through properties the
network has by construction: determinism, sample independence in eval mode, linearity of the backward in the
upstream gradient, invariance to lane padding and to LiDAR point order / out-of-range points, and the checkpoint
round trip through the flat parameter layout."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B = 32


@pytest.fixture(scope="module")
def rig():
    import bench
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    torch.manual_seed(42)
    net = MMFN(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0), DEV)
    inp, gt = bench.synth_inputs(B, torch.device(DEV), seed=42)
    eng = net._engine_for()
    net.train()
    eng.forward(inp, True, gt)  # one training forward gives the BN running stats a sane value for the eval checks
    torch.cuda.synchronize()
    return net, eng, inp, gt


def _slice(inp, idx):
    return {k: (v[idx].contiguous() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}


def test_eval_outputs_do_not_depend_on_batch_composition(rig):
    net, eng, inp, gt = rig
    net.eval()
    with torch.no_grad():
        full = eng.forward(inp, False, None)[0].clone()
        idx = torch.tensor([5, 17], device=DEV)
        pair = eng.forward(_slice(inp, idx), False, None)[0].clone()
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(DEV)
        shuffled = eng.forward(_slice(inp, perm), False, None)[0].clone()
    scale = full.abs().max().item()
    assert (full[idx] - pair).abs().max().item() <= 1e-5 * max(1.0, scale)
    assert (full[perm] - shuffled).abs().max().item() <= 1e-5 * max(1.0, scale)


def test_training_step_is_deterministic_and_backward_is_linear(rig):
    net, eng, inp, gt = rig
    net.train()
    L = net._layout
    snap = (L.params.clone(), L.buffers_flat.clone(), L.counters_flat.clone())

    def grads(gscale):
        L.params.copy_(snap[0]); L.buffers_flat.copy_(snap[1]); L.counters_flat.copy_(snap[2])
        _, loss = eng.forward(inp, True, gt)
        eng.backward(None, gscale)
        torch.cuda.synchronize()
        return loss.clone(), L.grads[:L.tail].clone()

    l1, g1 = grads(1.0)
    l2, g2 = grads(1.0)
    assert torch.equal(l1, l2) and torch.equal(g1, g2)          # split-K and reductions are order-fixed
    _, g4 = grads(4.0)
    assert torch.equal(g4, g1 * 4.0)                             # power-of-two scaling is exact in fp32
    assert torch.isfinite(g1).all() and g1.abs().max().item() > 0
    L.params.copy_(snap[0]); L.buffers_flat.copy_(snap[1]); L.counters_flat.copy_(snap[2])


def test_lane_padding_and_lidar_point_order_do_not_matter(rig):
    net, eng, inp, gt = rig
    net.eval()
    with torch.no_grad():
        ref = eng.forward(inp, False, None)[0].clone()
        wide = dict(inp)
        pad = torch.zeros(B, 96, 10, 5, device=DEV)
        pad[:, :64] = inp["lane"]
        wide["lane"] = pad                                        # 32 more padded lanes, same lane_num
        got = eng.forward(wide, False, None)[0].clone()
        assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
        # LiDAR: permuting the points and appending out-of-range ones leaves the integer histogram bit-identical
        from mmfn_amd import ops
        pts = inp["lidar_pts"]
        h0 = ops.lidar_splat(pts, torch.empty(B, 256, 256, 2, device=DEV)).clone()
        perm = torch.randperm(pts.shape[1], generator=torch.Generator().manual_seed(2)).to(DEV)
        junk = torch.tensor([[40.0, 0.0, 0.0, 0.0], [0.0, -24.001, 0.0, 0.0], [1e6, 1e6, -5.0, 0.0]], device=DEV)
        more = torch.cat([pts[:, perm], junk[None].expand(B, -1, -1)], 1).contiguous()
        h1 = ops.lidar_splat(more, torch.empty(B, 256, 256, 2, device=DEV))
        assert torch.equal(h0, h1)
        counts = (h0 * 5).round()
        assert torch.equal(counts / 5, h0) and counts.max().item() <= 5 and counts.sum().item() > 0


def test_checkpoint_round_trip_through_flat_layout(rig):
    net, eng, inp, gt = rig
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    other = MMFN(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0), DEV)
    other.load_state_dict(sd, strict=True)
    back = other.state_dict()
    assert list(back.keys()) == list(sd.keys()) and len(sd) == 1129
    assert sum(v.numel() for v in sd.values()) == 104_860_254  # 1129 keys, this many elements
    for k in sd:
        assert torch.equal(back[k], sd[k]), k
    net.eval(), other.eval()
    with torch.no_grad():
        a = eng.forward(inp, False, None)[0].clone()
        b = other._engine_for().forward(inp, False, None)[0].clone()
    assert torch.equal(a, b)


def test_adamw_is_a_no_op_without_gradient_or_decay(rig):
    net, eng, inp, gt = rig
    L = net._layout
    p0, m0, v0, s0 = L.params.clone(), L.exp_avg.clone(), L.exp_avg_sq.clone(), eng.step_count.clone()
    L.grads.zero_(); L.exp_avg.zero_(); L.exp_avg_sq.zero_()
    eng.optimizer_step(lr=1e-3, weight_decay=0.0)
    torch.cuda.synchronize()
    assert torch.equal(L.params, p0)
    L.exp_avg.copy_(m0); L.exp_avg_sq.copy_(v0); eng.step_count.copy_(s0)
