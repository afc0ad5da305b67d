"""GPU parity of every non-GEMM HIP kernel against plain torch fp32 (CPU) / the oracle."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, ref, tol=2e-5, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().double()
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, "%s max err %g vs scale %g" % (what, err, scale)


def _g(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------ BatchNorm
@pytest.mark.parametrize("M,C", [(2 * 16 * 16, 64), (3 * 8 * 8, 512), (1000, 128), (77, 256)])
def test_batchnorm_train_and_backward(M, C):
    from mmfn_amd import ops
    g = _g(M + C)
    x = torch.randn(M, C, generator=g) * 3 + 1.5
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    res = torch.randn(M, C, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    gy = torch.randn(M, C, generator=g)
    # reference through torch BatchNorm on NCHW view [1,C,M,1]
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train()
    xr = x.t().reshape(1, C, M, 1).clone().requires_grad_(True)
    rr = res.t().reshape(1, C, M, 1).clone().requires_grad_(True)
    y_ref = torch.relu(bn(xr) + rr)
    y_ref.backward(gy.t().reshape(1, C, M, 1))

    xd, wd, bd, resd, gyd = (t.to(DEV) for t in (x, w, b, res, gy))
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_train_stats(xd, mean, rstd, rmd, rvd, nbt)
    y = ops.bn_apply(xd, torch.empty_like(xd), mean, rstd, wd, bd, True, res=resd)
    _close(y, y_ref.reshape(C, M).t(), 1e-5, "bn fwd")
    _close(rmd, bn.running_mean, 1e-6, "running_mean")
    _close(rvd, bn.running_var, 1e-6, "running_var")
    assert int(nbt.item()) == 1
    dx, ge = torch.empty_like(xd), torch.empty_like(xd)
    dw, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_bwd(gyd, y, xd, mean, rstd, wd, dx, dw, db, ge_out=ge)
    _close(dx, xr.grad.reshape(C, M).t(), 2e-5, "bn dx")
    _close(ge, rr.grad.reshape(C, M).t(), 1e-6, "bn residual grad")
    _close(dw, bn.weight.grad, 2e-5, "bn dweight")
    _close(db, bn.bias.grad, 2e-5, "bn dbias")
    # eval mode
    bn.eval()
    ops.bn_eval_prepare(rmd, rvd, mean, rstd)
    y2 = ops.bn_apply(xd, torch.empty_like(xd), mean, rstd, wd, bd, False)
    _close(y2, bn(x.t().reshape(1, C, M, 1)).reshape(C, M).t(), 1e-5, "bn eval")


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,C,act", [(384, 64, 0), (100, 128, 1), (6144, 512, 0), (37, 256, 2), (18, 64, 2)])
def test_layernorm(M, C, act):
    from mmfn_amd import ops
    g = _g(M * 3 + C + act)
    x = (torch.randn(M, C, generator=g) * 2 + 0.3)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.2
    gy, dres = torch.randn(M, C, generator=g), torch.randn(M, C, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.layer_norm(xr, (C,), wr, br, 1e-5)
    y_ref = [y_ref, torch.relu(y_ref), F.gelu(y_ref)][act]
    y_ref.backward(gy)
    xd, wd, bd, gyd, dresd = (t.to(DEV) for t in (x, w, b, gy, dres))
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    y = ops.layernorm_fwd(xd, wd, bd, torch.empty_like(xd), mean, rstd, act)
    _close(y, y_ref, 1e-5, "ln fwd")
    dx, dw, db = torch.empty_like(xd), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.layernorm_bwd(gyd, xd, wd, bd, mean, rstd, dx, dw, db, act, dres=dresd)
    _close(dx, xr.grad + dres, 2e-5, "ln dx")
    _close(dw, wr.grad, 2e-5, "ln dw")
    _close(db, br.grad, 2e-5, "ln db")
    # optional second output: dx with the residual-branch dropout mask applied == a separate dropout pass over dx, bit for bit
    state = torch.tensor([1234, 7], dtype=torch.int64, device=DEV)
    dx2, dxd = torch.empty_like(xd), torch.empty_like(xd)
    ops.layernorm_bwd(gyd, xd, wd, bd, mean, rstd, dx2, dw, db, act, dres=dresd, dx_dropped=dxd, drop_p=0.1, rng_state=state,
                      rng_stream=42)
    assert torch.equal(dx2, dx)
    ref = ops.dropout_apply(dx, torch.empty_like(dx), 0.1, state, 42)
    assert torch.equal(dxd, ref) and (dxd == 0).float().mean().item() > 0.05
    # optional third output: column sums of the dropped copy (of dx itself without one) = the next Linear's bias gradient
    cs = torch.empty(C, device=DEV)
    ops.layernorm_bwd(gyd, xd, wd, bd, mean, rstd, dx2, dw, db, act, dres=dresd, dx_dropped=dxd, drop_p=0.1, rng_state=state,
                      rng_stream=42, dx_colsum=cs)
    _close(cs, dxd.double().sum(0).float().cpu(), 2e-5, "ln dx colsum (dropped)")
    _close(dw, wr.grad, 2e-5, "ln dw with the third partial row")
    ops.layernorm_bwd(gyd, xd, wd, bd, mean, rstd, dx2, dw, db, act, dres=dresd, dx_colsum=cs)
    _close(cs, dx.double().sum(0).float().cpu(), 2e-5, "ln dx colsum")


def test_layernorm_backward_finalize_batched_equals_one_by_one():
    """mmfn_layernorm_bwd_finalize_batched_f32: the row reductions of several LayerNorm backward passes of one shape in one launch
    (a transformer's 17 in the bf16 mode) - bit for bit what the single launches write, with and without the column-sum row."""
    from mmfn_amd import ops
    M, C = 768, 128
    g = _g(11)
    rows = ops.layernorm_bwd_rows(M)
    outs_a, outs_b, entries = [], [], []
    for k in range(5):
        x = (torch.randn(M, C, generator=g) * 1.5).to(DEV)
        gy = torch.randn(M, C, generator=g).to(DEV)
        w, b = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        ops.layernorm_fwd(x, w, b, torch.empty_like(x), mean, rstd, 0)
        want = k % 2 == 0
        part = torch.empty(rows, 3 if want else 2, C, device=DEV)
        ops.layernorm_bwd_partial(gy, x, w, b, mean, rstd, torch.empty_like(x), part, 0, want_colsum=want)
        a = [torch.empty(C, device=DEV) for _ in range(3 if want else 2)]
        bb = [torch.full((C,), float("nan"), device=DEV) for _ in range(3 if want else 2)]
        ops.layernorm_bwd_finalize(part, rows, C, a[0], a[1], a[2] if want else None)
        entries.append((part, bb[0], bb[1], bb[2] if want else None))
        outs_a.append(a); outs_b.append(bb)
    table = ops.layernorm_finalize_table(entries, DEV)
    ops.layernorm_bwd_finalize_batched(table, len(entries), rows, C)
    torch.cuda.synchronize()
    for a, bb in zip(outs_a, outs_b):
        for u, v in zip(a, bb):
            assert torch.equal(u, v)


def test_colsum():
    from mmfn_amd import ops
    x = torch.randn(777, 300, generator=_g(1))
    _close(ops.colsum(x.to(DEV), torch.empty(300, device=DEV)), x.sum(0), 1e-5)
    x = torch.randn(32, 70000, generator=_g(2))
    _close(ops.colsum(x.to(DEV), torch.empty(70000, device=DEV)), x.sum(0), 1e-5)


# ------------------------------------------------------------------ pooling family
def test_maxpool_ties_and_backward():
    from mmfn_amd import ops
    g = _g(3)
    x = torch.relu(torch.randn(2, 64, 18, 22, generator=g))  # many exact-zero ties, like post-ReLU maps
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    B, H, W, C = xd.shape
    y = torch.empty(B, y_ref.shape[2], y_ref.shape[3], C, device=DEV)
    idx = torch.empty(y.shape, dtype=torch.uint8, device=DEV)
    ops.maxpool_fwd(xd, y, idx)
    assert torch.equal(y.cpu().permute(0, 3, 1, 2), y_ref.detach())
    gx = ops.maxpool_bwd(gy.permute(0, 2, 3, 1).contiguous().to(DEV), idx, torch.empty_like(xd))
    _close(gx.permute(0, 3, 1, 2), xr.grad, 1e-6, "maxpool bwd")


@pytest.mark.parametrize("S,C", [(64, 64), (32, 128), (16, 256), (8, 512)])
def test_tokens_and_upsample(S, C):
    from mmfn_amd import ops
    g = _g(S + C)
    B, n = 2, 3
    feats = [torch.randn(B, C, S, S, generator=g) for _ in range(n)]
    pos = torch.randn(1, n * 64, C, generator=g)
    vw, vb, vel = torch.randn(C, 1, generator=g), torch.randn(C, generator=g), torch.rand(B, generator=g) * 8
    fr = [f.clone().requires_grad_(True) for f in feats]
    posr, vwr, vbr = pos.clone().requires_grad_(True), vw.clone().requires_grad_(True), vb.clone().requires_grad_(True)
    pooled = [F.adaptive_avg_pool2d(f, (8, 8)) for f in fr]
    tok_ref = torch.cat([p.flatten(2).transpose(1, 2) for p in pooled], 1)
    tok_ref = posr + tok_ref + F.linear(vel.unsqueeze(1), vwr, vbr).unsqueeze(1)
    gt = torch.randn(tok_ref.shape, generator=g)
    tok_ref.backward(gt)
    fd = [f.permute(0, 2, 3, 1).contiguous().to(DEV) for f in feats]
    tok = ops.tokens_fwd(fd, pos[0].contiguous().to(DEV), vw[:, 0].contiguous().to(DEV), vb.to(DEV), vel.to(DEV),
                         torch.empty(B, n * 64, C, device=DEV))
    _close(tok, tok_ref, 1e-5, "tokens fwd")
    gtd = gt.to(DEV).clone()
    dpos, dvw, dvb = torch.empty(n * 64, C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.tokens_bwd(gtd, vel.to(DEV), dpos, dvw, dvb)
    _close(dpos, posr.grad[0], 1e-5, "dpos")
    _close(dvw, vwr.grad[:, 0], 2e-5, "dvel_w")
    _close(dvb, vbr.grad, 2e-5, "dvel_b")
    # upsample-add forward, adjoint and the pooled-grad broadcast for modality m
    for m in range(n):
        t = torch.randn(B, n * 64, C, generator=g)
        tr = t.clone().requires_grad_(True)
        fm = feats[m].clone().requires_grad_(True)
        grid = tr[:, m * 64:(m + 1) * 64].view(B, 8, 8, C).permute(0, 3, 1, 2)
        up = grid if S == 8 else F.interpolate(grid, scale_factor=S // 8, mode="bilinear", align_corners=True)
        out_ref = fm + up
        G = torch.randn(out_ref.shape, generator=g)
        out_ref.backward(G)
        out = ops.upsample_add_fwd(fd[m], t.to(DEV), torch.empty_like(fd[m]), m)
        _close(out.permute(0, 3, 1, 2), out_ref, 1e-5, "upsample fwd")
        Gd = G.permute(0, 2, 3, 1).contiguous().to(DEV)
        gtok = torch.zeros(B, n * 64, C, device=DEV)
        ops.upsample_adj(Gd, gtok, m)
        _close(gtok[:, m * 64:(m + 1) * 64], tr.grad[:, m * 64:(m + 1) * 64], 2e-5, "upsample adjoint")
        # dF = G + avgpool adjoint of the token gradient
        dF = ops.pool_bcast_add(Gd, gt.to(DEV), torch.empty_like(Gd), m)
        _close(dF.permute(0, 3, 1, 2), G + fr[m].grad, 1e-5, "pool bcast add")


@pytest.mark.parametrize("S,C", [(64, 64), (16, 256), (8, 512)])
@pytest.mark.parametrize("frames", [[2, 2, 2], [2, 1, 1], [3, 1, 2]])
def test_tokens_and_upsample_with_several_frames(S, C, frames):
    """seq_len / n_views > 1: modality m holds frames[m] maps per sample, each its own 64-token group
    (GPT.forward, model_img.py:211-246); references built with the reference's view / cat / slice recipe."""
    from mmfn_amd import ops
    g = _g(S + C + sum(frames))
    B, n, ng = 2, len(frames), sum(frames)
    base = [sum(frames[:m]) for m in range(n)]
    feats = [torch.randn(B * f, C, S, S, generator=g) for f in frames]
    pos = torch.randn(1, ng * 64, C, generator=g)
    vw, vb, vel = torch.randn(C, 1, generator=g), torch.randn(C, generator=g), torch.rand(B, generator=g) * 8
    fr = [f.clone().requires_grad_(True) for f in feats]
    pooled = [F.adaptive_avg_pool2d(f, (8, 8)).view(B, k, C, 8, 8) for f, k in zip(fr, frames)]
    tok_ref = torch.cat(pooled, dim=1).permute(0, 1, 3, 4, 2).contiguous().view(B, -1, C)
    tok_ref = pos + tok_ref + F.linear(vel.unsqueeze(1), vw, vb).unsqueeze(1)
    gt = torch.randn(tok_ref.shape, generator=g)
    tok_ref.backward(gt)
    fd = [f.permute(0, 2, 3, 1).contiguous().to(DEV) for f in feats]
    tok = ops.tokens_fwd(fd, pos[0].contiguous().to(DEV), vw[:, 0].contiguous().to(DEV), vb.to(DEV), vel.to(DEV),
                         torch.empty(B, ng * 64, C, device=DEV), frames=frames)
    _close(tok, tok_ref, 1e-5, "tokens fwd")
    with pytest.raises(ValueError):
        ops.tokens_fwd(fd, pos[0].contiguous().to(DEV), vw[:, 0].contiguous().to(DEV), vb.to(DEV), vel.to(DEV),
                       torch.empty(B, ng * 64, C, device=DEV), frames=[f + 1 for f in frames])
    for m in range(n):
        k = frames[m]
        t = torch.randn(B, ng * 64, C, generator=g)
        tr = t.clone().requires_grad_(True)
        fm = feats[m].clone().requires_grad_(True)
        x = tr.view(B, ng, 8, 8, C).permute(0, 1, 4, 2, 3)
        grid = x[:, base[m]:base[m] + k].contiguous().view(B * k, C, 8, 8)
        up = grid if S == 8 else F.interpolate(grid, scale_factor=S // 8, mode="bilinear", align_corners=True)
        out_ref = fm + up
        G = torch.randn(out_ref.shape, generator=g)
        out_ref.backward(G)
        out = ops.upsample_add_fwd(fd[m], t.to(DEV), torch.empty_like(fd[m]), base[m], k)
        _close(out.permute(0, 3, 1, 2), out_ref, 1e-5, "upsample fwd")
        Gd = G.permute(0, 2, 3, 1).contiguous().to(DEV)
        gtok = torch.zeros(B, ng * 64, C, device=DEV)
        ops.upsample_adj(Gd, gtok, base[m], k)
        _close(gtok, tr.grad, 2e-5, "upsample adjoint")   # the other modalities' groups stay zero
        dF = ops.pool_bcast_add(Gd, gt.to(DEV), torch.empty_like(Gd), base[m], k)
        _close(dF.permute(0, 3, 1, 2), G + fr[m].grad, 1e-5, "pool bcast add")
    # the final global-average-pool + sum over every frame of every modality (model_img.py:410-423)
    f8 = [torch.randn(B * k, 512, 8, 8, generator=g).requires_grad_(True) for k in frames]
    ref = torch.cat([f.mean(dim=(2, 3)).view(B, k, -1) for f, k in zip(f8, frames)], dim=1).sum(dim=1)
    gg = torch.randn(B, 512, generator=g)
    ref.backward(gg)
    f8d = [f.detach().permute(0, 2, 3, 1).contiguous().to(DEV) for f in f8]
    _close(ops.gap_sum_fwd(f8d, torch.empty(B, 512, device=DEV), frames=frames), ref, 1e-5)
    outs = [torch.full_like(f, float("nan")) for f in f8d]
    ops.gap_sum_bwd(gg.to(DEV), outs, frames=frames)
    for o, f in zip(outs, f8):
        _close(o.permute(0, 3, 1, 2), f.grad, 1e-6)


@pytest.mark.parametrize("B,H,W,C,relu,with_res,want_y", [(2, 16, 16, 64, True, False, False), (2, 16, 16, 64, True, True, True),
                                                         (3, 8, 8, 512, True, True, True), (1, 64, 64, 64, True, False, False),
                                                         (2, 32, 32, 128, False, True, True), (2, 8, 8, 256, False, False, False)])
def test_winograd_input_transform_applies_the_producers_batchnorm(B, H, W, C, relu, with_res, want_y):
    """mmfn_wino_input_bn_f32 (BatchNorm apply + residual + ReLU of the producing layer inside the consumer's F(4x4) input
    transform, model_vec.py:509-593 BasicBlock chain) == mmfn_bn_apply_f32 followed by mmfn_wino_input_f32, BIT FOR BIT: the
    transformed input, and the activation tensor when it is requested; nothing is written when it is not."""
    from mmfn_amd import ops
    from mmfn_amd.ops import _call, ptr, stream
    g = _g(B * H + C)
    M = B * H * W
    co = (torch.randn(B, H, W, C, generator=g) * 2 + 0.3).to(DEV)
    res = torch.randn(B, H, W, C, generator=g).to(DEV) if with_res else None
    mean, rstd = (torch.randn(C, generator=g) * 0.2).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    w, b = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    y_ref = ops.bn_apply(co.view(M, C), torch.empty(M, C, device=DEV), mean, rstd, w, b, relu, res=None if res is None else res.view(M, C))
    T = B * (H // 4) * (W // 4)
    V_ref = torch.empty(36 * T * C, device=DEV)
    _call("mmfn_wino_input_f32", ptr(y_ref), ptr(V_ref), B, H, W, C, 4, stream())
    V = torch.full((36 * T * C,), float("nan"), device=DEV)
    y = torch.full((M, C), float("nan"), device=DEV) if want_y else None
    _call("mmfn_wino_input_bn_f32", ptr(co), ptr(res), ptr(mean), ptr(rstd), ptr(w), ptr(b), 1 if relu else 0, ptr(y), ptr(V), B, H, W, C,
          stream())
    assert torch.equal(V, V_ref)
    if want_y:
        assert torch.equal(y, y_ref)
    # against torch: relu(bn(x) + res) then the same transform through the convolution it feeds
    ref = (co - mean) * rstd * w + b
    if res is not None:
        ref = ref + res
    if relu:
        ref = torch.relu(ref)
    _close(y_ref.view(B, H, W, C), ref.cpu(), 1e-5, "bn apply")


@pytest.mark.parametrize("M,C,HW", [(2 * 16 * 16, 64, 16), (3 * 8 * 8, 512, 8), (2 * 32 * 32, 128, 32)])
def test_batchnorm_backward_recomputes_the_relu_mask_of_an_unwritten_output(M, C, HW):
    """The backward kernels of a BatchNorm + ReLU whose output never reached HBM (applied inside the next convolution's input
    transform): the mask recomputed from the convolution output (relu_bias / relu_wb) gives bit-identical results to the mask read
    from the written activation - reductions, the apply pass, and the Winograd output-gradient transform."""
    from mmfn_amd import ops
    from mmfn_amd.ops import _call, ptr, stream
    g = _g(M + C + 1)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(DEV)
    w, b = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    mean, rstd = x.mean(0), (x.var(0, unbiased=False) + 1e-5).rsqrt()
    gy = torch.randn(M, C, generator=g).to(DEV)
    y = ops.bn_apply(x, torch.empty_like(x), mean, rstd, w, b, True)
    assert 0.2 < (y > 0).float().mean().item() < 0.8
    outs = []
    for ymask, rb in ((y, None), (None, b)):
        dx, ge = torch.empty_like(x), torch.empty_like(x)
        dw, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        ops.bn_bwd(gy, ymask, x, mean, rstd, w, dx, dw, db, ge_out=ge, relu_bias=rb)
        dw2, db2, means = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(2, C, device=DEV)
        ops.bn_bwd_reduce(gy, ymask, x, mean, rstd, dw2, db2, means, relu_wb=None if rb is None else (w, rb))
        B = M // (HW * HW)
        dMt = torch.empty(36 * B * (HW // 4) * (HW // 4) * C, device=DEV)
        ge2 = torch.empty_like(x)
        _call("mmfn_wino_outgrad_bn_f32", ptr(gy), ptr(ymask), ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(rb), ptr(means), ptr(ge2), ptr(dMt),
              B, HW, HW, C, stream())
        outs.append((dx, ge, dw, db, dw2, db2, means, dMt, ge2))
    for name, a, c in zip(("dx", "ge", "dweight", "dbias", "dweight (reduce)", "dbias (reduce)", "means", "dM", "ge (transform)"), *outs):
        assert torch.equal(a, c), (name, int((a != c).sum()), float((a - c).abs().max()))
    # and without either the mask is NOT applied (a caller that forgets relu_bias would silently train a different network)
    dxn = torch.empty_like(x)
    ops.bn_bwd(gy, None, x, mean, rstd, w, dxn, torch.empty(C, device=DEV), torch.empty(C, device=DEV))
    assert not torch.equal(dxn, outs[0][0])


def test_gap_and_transpose():
    from mmfn_amd import ops
    g = _g(9)
    feats = [torch.randn(3, 512, 8, 8, generator=g) for _ in range(3)]
    ref = sum(f.mean(dim=(2, 3)) for f in feats)
    fd = [f.permute(0, 2, 3, 1).contiguous().to(DEV) for f in feats]
    _close(ops.gap_sum_fwd(fd, torch.empty(3, 512, device=DEV)), ref, 1e-5)
    gg = torch.randn(3, 512, generator=g)
    outs = [torch.empty_like(f) for f in fd]
    ops.gap_sum_bwd(gg.to(DEV), outs)
    for o in outs:
        _close(o, (gg / 64)[:, None, None, :].expand(3, 8, 8, 512), 1e-6)
    x = torch.randn(2, 70, 4096, generator=g)
    _close(ops.transpose(x.to(DEV), torch.empty(2, 4096, 70, device=DEV), 2, 70, 4096), x.transpose(1, 2), 0.0)


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("T,NH,HS", [(192, 4, 16), (192, 4, 32), (192, 4, 64), (192, 4, 128), (256, 4, 128),
                                     (128, 2, 64), (128, 4, 128), (64, 2, 64), (64, 3, 32), (9, 2, 64), (50, 2, 64),
                                     (384, 4, 16), (384, 4, 32), (320, 4, 64), (384, 4, 128),     # seq_len / n_views > 1: tile kernels ...
                                     (256, 4, 16), (256, 4, 32), (256, 4, 64), (320, 4, 16)])    # ... and the workgroup kernels where they fit (T = 256: four per (sample, head))
@pytest.mark.parametrize("masked", [False, True])
def test_attention(T, NH, HS, masked):
    from mmfn_amd import ops
    if masked and T > 64:
        pytest.skip("key mask only used by the lane attention")
    g = _g(T * 5 + HS)
    B, C = 3, NH * HS
    qkv = torch.randn(B, T, 3 * C, generator=g)
    dO = torch.randn(B, T, C, generator=g)
    scale = 1.0 / math.sqrt(HS)
    kv_len = torch.tensor([T, max(1, T // 3), max(1, T - 1)], dtype=torch.int32) if masked else None
    qr = qkv.clone().requires_grad_(True)
    q, k, v = (t.view(B, T, NH, HS).transpose(1, 2) for t in qr.chunk(3, dim=-1))
    att = (q @ k.transpose(-1, -2)) * scale
    if masked:
        keep = (torch.arange(T)[None, :] < kv_len[:, None]).view(B, 1, 1, T)
        att = att.masked_fill(~keep, -1e9)
    o_ref = (torch.softmax(att, -1) @ v).transpose(1, 2).reshape(B, T, C)
    o_ref.backward(dO)
    qd = qkv.to(DEV).view(B * T, 3 * C)
    o = torch.empty(B * T, C, device=DEV)
    lse = torch.empty(B, NH, T, device=DEV)
    kvd = kv_len.to(DEV) if masked else None
    ops.attention_fwd(qd, qd[:, C:], qd[:, 2 * C:], 3 * C, o, C, lse, B, T, NH, HS, scale, kv_len=kvd)
    _close(o.view(B, T, C), o_ref, 2e-5, "attn fwd")
    dqkv = torch.zeros(B * T, 3 * C, device=DEV)
    delta = torch.empty(B, NH, T, device=DEV)
    ops.attention_bwd(qd, qd[:, C:], qd[:, 2 * C:], 3 * C, o, dO.to(DEV).view(B * T, C), C, lse, delta, dqkv, dqkv[:, C:],
                      dqkv[:, 2 * C:], 3 * C, B, T, NH, HS, scale, kv_len=kvd)
    _close(dqkv.view(B, T, 3 * C), qr.grad, 5e-5, "attn bwd")


@pytest.mark.parametrize("T", [48, 64])   # 48: tile kernels (attention.hip); 64: workgroup-per-half kernels (attention_wg.hip)
def test_attention_sample_without_keys_is_uniform_and_finite(T):
    """kv_len[b] == 0 (a sample with zero lanes): the reference's masked_fill(-1e9) + softmax gives uniform attention over
    the padded keys and a zero score gradient (model_vec.py:315-317); the kernels must do the same instead of 0/0 = NaN."""
    from mmfn_amd import ops
    B, NH, HS = 3, 2, 64
    C = NH * HS
    g = _g(77)
    qkv = torch.randn(B, T, 3 * C, generator=g)
    dO = torch.randn(B, T, C, generator=g)
    scale = HS ** -0.5
    kv_len = torch.tensor([0, 5, 0], dtype=torch.int32)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = (t.view(B, T, NH, HS).transpose(1, 2) for t in qr.chunk(3, dim=-1))
    keep = (torch.arange(T)[None, :] < kv_len[:, None]).view(B, 1, 1, T)
    att = ((q @ k.transpose(-1, -2)) * scale).masked_fill(~keep, -1e9)
    o_ref = (torch.softmax(att, -1) @ v).transpose(1, 2).reshape(B, T, C)
    o_ref.backward(dO)
    qd = qkv.to(DEV).view(B * T, 3 * C)
    o = torch.empty(B * T, C, device=DEV)
    lse = torch.empty(B, NH, T, device=DEV)
    ops.attention_fwd(qd, qd[:, C:], qd[:, 2 * C:], 3 * C, o, C, lse, B, T, NH, HS, scale, kv_len=kv_len.to(DEV))
    assert torch.isfinite(o).all()
    _close(o.view(B, T, C), o_ref, 2e-5, "attn fwd, empty key set")
    dqkv = torch.zeros(B * T, 3 * C, device=DEV)
    delta = torch.empty(B, NH, T, device=DEV)
    ops.attention_bwd(qd, qd[:, C:], qd[:, 2 * C:], 3 * C, o, dO.to(DEV).view(B * T, C), C, lse, delta, dqkv, dqkv[:, C:],
                      dqkv[:, 2 * C:], 3 * C, B, T, NH, HS, scale, kv_len=kv_len.to(DEV))
    assert torch.isfinite(dqkv).all()
    _close(dqkv.view(B, T, 3 * C), qr.grad, 5e-5, "attn bwd, empty key set")
    assert dqkv.view(B, T, 3 * C)[0, :, :2 * C].abs().max().item() == 0.0   # constant scores: no gradient to q, k


def test_attention_dropout_consistency():
    """Dropout mask is a pure function of (state, stream, index): fwd and bwd see the same mask."""
    from mmfn_amd import ops
    B, T, NH, HS = 2, 192, 4, 32
    C = NH * HS
    g = _g(11)
    qkv = torch.randn(B * T, 3 * C, generator=g).to(DEV)
    dO = torch.randn(B * T, C, generator=g).to(DEV)
    state = torch.tensor([99, 3], dtype=torch.int64, device=DEV)
    scale = 1.0 / math.sqrt(HS)
    p = 0.1

    def run(x):
        o = torch.empty(B * T, C, device=DEV)
        lse = torch.empty(B, NH, T, device=DEV)
        ops.attention_fwd(x, x[:, C:], x[:, 2 * C:], 3 * C, o, C, lse, B, T, NH, HS, scale, drop_p=p, rng_state=state,
                          rng_stream=5)
        return o, lse

    o, lse = run(qkv)
    o2, _ = run(qkv)
    assert torch.equal(o, o2)
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty(B, NH, T, device=DEV)
    ops.attention_bwd(qkv, qkv[:, C:], qkv[:, 2 * C:], 3 * C, o, dO, C, lse, delta, dqkv, dqkv[:, C:], dqkv[:, 2 * C:],
                      3 * C, B, T, NH, HS, scale, drop_p=p, rng_state=state, rng_stream=5)
    # directional finite difference of <o, dO> against the analytic gradient
    d = torch.randn(qkv.shape, generator=_g(12)).to(DEV)
    eps = 1e-2
    op, _ = run(qkv + eps * d)
    om, _ = run(qkv - eps * d)
    fd = ((op.double() - om.double()) * dO.double()).sum().item() / (2 * eps)
    an = (dqkv.double() * d.double()).sum().item()
    assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), (fd, an)
    # the mask drops ~p of the probabilities: outputs differ from the no-dropout ones
    o3 = torch.empty_like(o)
    ops.attention_fwd(qkv, qkv[:, C:], qkv[:, 2 * C:], 3 * C, o3, C, lse, B, T, NH, HS, scale)
    assert not torch.allclose(o, o3)


# ------------------------------------------------------------------ GRU head
def test_gru_head():
    from mmfn_amd import ops
    g = _g(21)
    B, steps = 5, 4
    gru = torch.nn.GRUCell(2, 64)
    out = torch.nn.Linear(64, 2)
    z0 = torch.randn(B, 64, generator=g).requires_grad_(True)
    target = torch.randn(B, 2, generator=g) * 5
    gt = torch.randn(B, steps, 2, generator=g) * 3
    x = torch.zeros(B, 2)
    h = z0
    wps = []
    for _ in range(steps):
        h = gru(x + target, h)
        x = x + out(h)
        wps.append(x)
    pred_ref = torch.stack(wps, 1)
    loss_ref = F.l1_loss(pred_ref, gt, reduction="none").mean()
    loss_ref.backward()
    P = {k: v.detach().to(DEV).contiguous() for k, v in dict(w_ih=gru.weight_ih, w_hh=gru.weight_hh, b_ih=gru.bias_ih,
                                                             b_hh=gru.bias_hh, w_out=out.weight, b_out=out.bias).items()}
    pred = torch.empty(B, steps, 2, device=DEV)
    hs = torch.empty(B, steps + 1, 64, device=DEV)
    gates = torch.empty(B, steps, 4, 64, device=DEV)
    xin = torch.empty(B, steps, 2, device=DEV)
    lt, loss = torch.empty(B, device=DEV), torch.empty(1, device=DEV)
    ops.gru_head_fwd(z0.detach().to(DEV), target.to(DEV), P["w_ih"], P["w_hh"], P["b_ih"], P["b_hh"], P["w_out"], P["b_out"],
                     gt.to(DEV), pred, hs, gates, xin, lt, loss, steps)
    _close(pred, pred_ref, 1e-5, "gru pred")
    assert abs(loss.item() - loss_ref.item()) < 1e-6
    from mmfn_amd._lib import lib
    npart = lib().mmfn_gru_head_part_floats()
    part = torch.empty(B, npart, device=DEV)
    dz0 = torch.empty(B, 64, device=DEV)
    ops.gru_head_bwd(pred, gt.to(DEV), None, 1.0 / (B * steps * 2), P["w_ih"], P["w_hh"], P["w_out"], hs, gates, xin, dz0,
                     part, steps)
    _close(dz0, z0.grad, 2e-5, "dz0")
    tot = part.sum(0).cpu()
    offs = [("w_ih", 384, gru.weight_ih), ("w_hh", 12288, gru.weight_hh), ("b_ih", 192, gru.bias_ih),
            ("b_hh", 192, gru.bias_hh), ("w_out", 128, out.weight), ("b_out", 2, out.bias)]
    o = 0
    for name, n, prm in offs:
        _close(tot[o:o + n], prm.grad.flatten(), 5e-5, name)
        o += n


# ------------------------------------------------------------------ AdamW
def test_adamw_matches_torch():
    from mmfn_amd import ops
    g = _g(31)
    n = 10007
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) for _ in range(3)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-4)
    pd = torch.zeros(n + 5, device=DEV)[:n]
    pd.copy_(p0)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    for gr in grads:
        ref.grad = gr.clone()
        opt.step()
        ops.step_advance(step)
        ops.adamw(pd, gr.to(DEV), m, v, step)
    assert (pd.cpu() - ref.detach()).abs().max().item() < 1e-6  # a few fp32 ulps at |p| ~ 3


# ------------------------------------------------------------------ ingest
def test_ingest_and_splat(golden_dir):
    from mmfn_amd import ops
    from oracle import fixtures, preprocess
    from oracle.model import normalize_imagenet
    batch = fixtures.synthetic_batch(2, seed=5)
    rgb = batch["rgb_u8"]
    ref = normalize_imagenet(torch.from_numpy(np.stack([preprocess.crop_chw(im) for im in rgb.numpy()]).copy()).float())
    out = ops.ingest_rgb_u8(rgb.to(DEV), torch.empty(2, 256, 256, 3, device=DEV))
    _close(out.permute(0, 3, 1, 2), ref, 1e-6, "rgb ingest")
    # NCHW f32 module-boundary path with the same normalisation
    raw = torch.from_numpy(np.stack([preprocess.crop_chw(im) for im in rgb.numpy()]).copy()).float()
    mean = torch.tensor([0.485, 0.456, 0.406])
    inv = torch.tensor([1 / 0.229, 1 / 0.224, 1 / 0.225])
    out2 = ops.nchw_to_nhwc(raw.to(DEV), torch.empty(2, 256, 256, 3, device=DEV), mean.to(DEV), inv.to(DEV))
    _close(out2.permute(0, 3, 1, 2), ref, 1e-6, "nchw ingest")
    # LiDAR splat: bit-exact against the oracle histogram and the reference-generated edge cases
    pts = batch["lidar_pts"]
    bev_ref = np.stack([preprocess.lidar_histogram(p[:, :3].numpy().astype(np.float64)) for p in pts])
    bev = ops.lidar_splat(pts.to(DEV), torch.empty(2, 256, 256, 2, device=DEV))
    assert np.array_equal(bev.cpu().permute(0, 3, 1, 2).numpy(), bev_ref)
    gold = np.load(os.path.join(golden_dir, "preprocess.npz"))
    for key_p, key_o in (("hist_pts", "hist_out"), ("hist_rand_pts", "hist_rand_out")):
        p = torch.from_numpy(gold[key_p].astype(np.float32))[None].contiguous()
        got = ops.lidar_splat(p.to(DEV), torch.empty(1, 256, 256, 2, device=DEV)).cpu().permute(0, 3, 1, 2).numpy()[0]
        want = preprocess.lidar_histogram(gold[key_p].astype(np.float32).astype(np.float64))
        assert np.array_equal(got, want)
        if key_p == "hist_rand_pts":
            assert np.array_equal(got, gold[key_o])


def test_lane_to_vector_and_polyline_pool():
    from mmfn_amd import ops
    from oracle.model import _VectornetEncoder
    g = _g(41)
    lane = torch.randn(2, 7, 10, 5, generator=g)
    vec = ops.lane_to_vector(lane.to(DEV), torch.empty(2 * 7 * 9, 7, device=DEV))
    _close(vec.view(2, 7, 9, 7), _VectornetEncoder.lane_to_vector(lane), 0.0)
    R, V, H = 37, 9, 64
    y = torch.relu(torch.randn(R, V, H, generator=g))  # ReLU output: ties at 0
    for last in (False, True):
        yr = y.clone().requires_grad_(True)
        pooled = yr.max(dim=-2, keepdim=True).values.expand_as(yr)
        out_ref = torch.cat([yr, pooled], -1)
        if last:
            out_ref = out_ref.max(dim=-2).values
        gout = torch.randn(out_ref.shape, generator=g)
        out_ref.backward(gout)
        out = torch.empty(out_ref.shape, device=DEV)
        arg = torch.empty(R, H, dtype=torch.uint8, device=DEV)
        ops.polyline_pool_fwd(y.to(DEV), out, arg, R, V, H, last)
        assert torch.equal(out.cpu(), out_ref.detach())
        gy = ops.polyline_pool_bwd(gout.to(DEV), arg, torch.empty(R, V, H, device=DEV), R, V, H, last)
        _close(gy, yr.grad, 1e-6, "polyline bwd")


@pytest.mark.parametrize("B,H,W,Cin,k,stride,pad,KP", [(2, 64, 64, 3, 7, 2, 3, 160), (3, 36, 52, 2, 7, 2, 3, 128), (1, 20, 20, 4, 3, 1, 1, 36),
                                                       (2, 16, 24, 1, 1, 1, 0, 4), (2, 64, 64, 3, 7, 2, 3, 192)])
@pytest.mark.parametrize("out", ["f32", "bf16"])
def test_im2col_small_matches_unfold(B, H, W, Cin, k, stride, pad, KP, out):
    """mmfn_im2col_small (the stems' im2col: tap table in LDS, reciprocal pixel decode, shifts where the output sides are powers of
    two and divisions where they are not) against F.unfold; columns K .. KP zero-filled."""
    from mmfn_amd import ops
    g = _g(B * 7 + H + Cin + k)
    x = torch.randn(B, H, W, Cin, generator=g)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    col = torch.full((B * OH * OW, KP), float("nan"), device=DEV, dtype=torch.bfloat16 if out == "bf16" else torch.float32)
    ops.im2col_small(x.to(DEV), col, k, k, stride, pad)
    u = F.unfold(x.permute(0, 3, 1, 2), k, padding=pad, stride=stride)            # [B, Cin*k*k, OH*OW], rows ordered (ci, kh, kw)
    u = u.view(B, Cin, k * k, OH * OW).permute(0, 3, 2, 1).reshape(B * OH * OW, k * k * Cin)   # -> (kh, kw, ci) fastest ci
    ref = torch.zeros(B * OH * OW, KP)
    ref[:, :k * k * Cin] = u
    if out == "bf16":
        ref = ref.bfloat16().float()
    assert torch.equal(col.float().cpu(), ref)


def test_counter_rng_masks_are_bernoulli_and_independent():
    """The dropout mask is a 32-bit mixer of (flat index ^ salt(seed, step, stream)) (csrc/common.h): keep rate 1 - p to sampling
    error, no structure along rows / columns, and masks of different streams, steps and seeds agree only as often as independent
    Bernoulli draws do (p^2 + (1 - p)^2) - a salt that failed to separate them would show as agreement 1."""
    from mmfn_amd import ops
    n, p = 1 << 22, 0.1
    ones = torch.ones(n, device=DEV)

    def mask(seed, step, stream):
        st = torch.tensor([seed, step], dtype=torch.int64, device=DEV)
        return ops.dropout_apply(ones, torch.empty_like(ones), p, st, stream) > 0

    m = mask(42, 7, 3)
    sd = math.sqrt(p * (1 - p) / n)
    assert abs(float(m.float().mean()) - (1 - p)) < 5 * sd
    assert torch.equal(m, mask(42, 7, 3))                                   # a pure function of (seed, step, stream, index)
    # rows / columns of a [2048, 2048] view: every row and column mean within 6 sigma of 1 - p, neighbours uncorrelated
    g = m.view(2048, 2048).float()
    sd_row = math.sqrt(p * (1 - p) / 2048)
    assert float((g.mean(0) - (1 - p)).abs().max()) < 6 * sd_row and float((g.mean(1) - (1 - p)).abs().max()) < 6 * sd_row
    for a, b in ((g[:, 1:], g[:, :-1]), (g[1:], g[:-1])):
        corr = float(((a - (1 - p)) * (b - (1 - p))).mean()) / (p * (1 - p))
        assert abs(corr) < 5 / math.sqrt(a.numel()), corr
    indep = p * p + (1 - p) * (1 - p)
    for other in (mask(42, 7, 4), mask(42, 8, 3), mask(43, 7, 3), mask(42, 7, 3 + (1 << 20))):
        agree = float((m == other).float().mean())
        assert abs(agree - indep) < 6 * math.sqrt(indep * (1 - indep) / n), agree
