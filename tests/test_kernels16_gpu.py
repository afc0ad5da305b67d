"""The *_bf16 entry points (bf16 activations in HBM) against their fp32 twins: the arithmetic is the same fp32 code (the kernels
are templates on the element type), so on bf16-exact inputs a bf16 output must be the round-to-nearest-even bf16 of the fp32
kernel's output - exactly, except where the two instantiations contract a multiply-add differently (an fp32 ulp that moves a
value across a bf16 rounding boundary: at most one bf16 ulp, on a vanishing fraction of the elements) - and every fp32 output
whose inputs are the same numbers (statistics, parameter gradients, lse) must be bit-identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _r(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(BF)


def _same(a16, a32):
    assert a16.dtype == BF
    want = a32.to(BF)
    if torch.equal(a16, want):
        return
    diff = (a16.float() - want.float()).abs()
    ulp = want.float().abs().clamp_min(1e-30) * 2.0 ** -7          # one bf16 ulp is at most 2^-7 of the value
    assert bool((diff <= ulp).all()), float((diff / ulp).max())
    assert float((diff > 0).float().mean()) <= 1e-3, float((diff > 0).float().mean())


def test_batchnorm_forward_backward():
    from mmfn_amd import ops
    M, C = 4096, 128
    x, res, g = _r(M, C, seed=1, scale=2.0), _r(M, C, seed=2), _r(M, C, seed=3)
    w, b = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    outs = []
    for dt in (torch.float32, BF):
        xx, rr, gg = x.to(dt), res.to(dt), g.to(dt)
        mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        rm, rv, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        ops.bn_train_stats(xx, mean, rstd, rm, rv, nbt)
        y = torch.empty(M, C, dtype=dt, device=DEV)
        ops.bn_apply(xx, y, mean, rstd, w, b, True, res=rr)
        dx, ge = torch.empty(M, C, dtype=dt, device=DEV), torch.empty(M, C, dtype=dt, device=DEV)
        dw, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        ops.bn_bwd(gg, y, xx, mean, rstd, w, dx, dw, db, ge_out=ge)
        outs.append((mean, rstd, rm, rv, y, dx, ge, dw, db))
    f, h = outs
    for i in (0, 1, 2, 3):
        assert torch.equal(f[i], h[i])
    _same(h[4], f[4])
    # the backward's mask uses the STORED y: identical here because relu output rounds to zero only where it is zero
    _same(h[6], f[6])
    _same(h[5], f[5])
    assert torch.equal(f[7], h[7]) and torch.equal(f[8], h[8])
    # stem form: fp32 convolution output, bf16 activations
    xx = x.float() * 1.0001   # not bf16-exact
    mean, rstd = f[0], f[1]
    y32, y16 = torch.empty(M, C, device=DEV), torch.empty(M, C, dtype=BF, device=DEV)
    ops.bn_apply(xx, y32, mean, rstd, w, b, True)
    ops.bn_apply(xx, y16, mean, rstd, w, b, True)
    _same(y16, y32)
    dx32, dx_mixed = torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV)
    dw, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_bwd(g.float(), y16.float(), xx, mean, rstd, w, dx32, dw, db)
    ops.bn_bwd(g, y16, xx, mean, rstd, w, dx_mixed, dw, db)
    assert torch.equal(dx32, dx_mixed)


@pytest.mark.parametrize("C", [64, 128, 256, 512])
def test_layernorm_forward_backward(C):
    from mmfn_amd import ops
    M = 1536
    x, g, dres = _r(M, C, seed=1), _r(M, C, seed=2), _r(M, C, seed=3)
    w, b = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    rng = torch.tensor([77, 3], dtype=torch.int64, device=DEV)
    res = []
    for dt in (torch.float32, BF):
        xx = x.to(dt)
        y, mean, rstd = torch.empty(M, C, dtype=dt, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        ops.layernorm_fwd(xx, w, b, y, mean, rstd, ops.ACT_RELU)
        dx, dxd = torch.empty(M, C, dtype=dt, device=DEV), torch.empty(M, C, dtype=dt, device=DEV)
        dw, db, cs = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        ops.layernorm_bwd(g.to(dt), xx, w, b, mean, rstd, dx, dw, db, ops.ACT_RELU, dres=dres.to(dt), dx_dropped=dxd, drop_p=0.1,
                          rng_state=rng, rng_stream=5, dx_colsum=cs)
        res.append((mean, rstd, y, dx, dxd, dw, db, cs))
    f, h = res
    assert torch.equal(f[0], h[0]) and torch.equal(f[1], h[1])
    _same(h[2], f[2])
    _same(h[3], f[3])
    assert torch.equal(f[5], h[5]) and torch.equal(f[6], h[6])
    # the dropped copy is the dropout of the fp32 dx rounded once (not of the rounded dx): compare against that
    keep = (f[4] != 0) | (f[3] == 0)
    _same(h[4], f[4])
    assert keep.float().mean() > 0.85


@pytest.mark.parametrize("C,drop_p", [(64, 0.0), (256, 0.1), (512, 0.0), (512, 0.1)])
def test_layernorm_over_the_fp32_residual_stream(C, drop_p):
    """bf16 mode inside the fusion transformers (model_vec.py:124-132 under torch.autocast: x + Linear(LN(x)) with x fp32): the
    LayerNorm reads the fp32 stream and writes the bf16 GEMM operand; its backward takes the bf16 operand gradient, adds the fp32
    stream gradient, writes dx in fp32 and the (dropped) bf16 copy the next GEMM reads.  Against the fp32 kernels on the same
    numbers: statistics, dx, parameter gradients bit-identical; y and the copy are their bf16 roundings; drop_p = 0 gives the
    plain copy."""
    from mmfn_amd import ops
    M = 1536
    g16 = _r(M, C, seed=2)
    gx = torch.Generator().manual_seed(4)
    x = (torch.randn(M, C, generator=gx) * 1.3 + 0.2).to(DEV)            # NOT bf16-exact: the stream is fp32
    dres = torch.randn(M, C, generator=gx).to(DEV)
    w, b = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    rng = torch.tensor([77, 3], dtype=torch.int64, device=DEV)
    out = []
    for mixed in (False, True):
        ydt = BF if mixed else torch.float32
        y, mean, rstd = torch.empty(M, C, dtype=ydt, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        ops.layernorm_fwd(x, w, b, y, mean, rstd)
        dx, dxd = torch.empty(M, C, device=DEV), torch.empty(M, C, dtype=ydt, device=DEV)
        dw, db, cs = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        g = g16 if mixed else g16.float()
        ops.layernorm_bwd(g, x, w, b, mean, rstd, dx, dw, db, dres=dres, dx_dropped=dxd, drop_p=drop_p,
                          rng_state=rng if drop_p > 0 else None, rng_stream=5, dx_colsum=cs)
        out.append((mean, rstd, y, dx, dxd, dw, db, cs))
    f, h = out
    assert torch.equal(f[0], h[0]) and torch.equal(f[1], h[1])
    _same(h[2], f[2])
    assert torch.equal(f[3], h[3]), float((f[3] - h[3]).abs().max())     # dx: the same fp32 arithmetic on the same numbers
    _same(h[4], f[4])
    assert torch.equal(f[5], h[5]) and torch.equal(f[6], h[6])
    if drop_p == 0.0:
        assert torch.equal(h[4], h[3].to(BF))
    # column sums: of the copy that LEAVES (the bf16 kernel sums the fp32 value before rounding, like the fp32 kernel)
    assert (f[7] - h[7]).abs().max().item() <= 1e-5 * max(1.0, f[7].abs().max().item())


def test_tokens_and_pool_adjoint_with_the_fp32_token_stream():
    """bf16 feature maps, fp32 token matrix / token gradient (the transformers' residual stream): mmfn_tokens_fwd_bf16(tok_is_f32)
    and mmfn_pool_bcast_add_bf16(gtok_is_f32) against the fp32 kernels on the same numbers; the bf16 GEMM's fp32 residual."""
    from mmfn_amd import ops, ops16
    B, S, C, T = 2, 32, 64, 192
    feats = [_r(B, S, S, C, seed=i) for i in range(3)]
    pos, vw, vb, vel = torch.randn(T, C, device=DEV), torch.randn(C, device=DEV), torch.randn(C, device=DEV), torch.rand(B, device=DEV)
    rng = torch.tensor([5, 1], dtype=torch.int64, device=DEV)
    t32 = ops.tokens_fwd([f.float() for f in feats], pos, vw, vb, vel, torch.empty(B, T, C, device=DEV), 0.1, rng, 3)
    tmix = ops.tokens_fwd(feats, pos, vw, vb, vel, torch.empty(B, T, C, device=DEV), 0.1, rng, 3)
    assert tmix.dtype == torch.float32 and torch.equal(tmix, t32)
    G = _r(B, S, S, C, seed=11)
    gtok = torch.randn(B, T, C, device=DEV)
    for m in range(3):
        d32 = ops.pool_bcast_add(G.float(), gtok, torch.empty(B, S, S, C, device=DEV), m)
        dmix = ops.pool_bcast_add(G, gtok, torch.empty(B, S, S, C, dtype=BF, device=DEV), m)
        _same(dmix, d32)
    # x1 = x + Linear(o): bf16 operands, fp32 residual in, fp32 sum out
    M, K, N = 384, 256, 256
    o, wt = _r(M, K, seed=21), _r(N, K, seed=22, scale=0.05)
    bias, res = torch.randn(N, device=DEV), torch.randn(M, N, device=DEV)
    got = ops16.linear_fwd(o, wt, bias, out=torch.empty(M, N, device=DEV), res=res, ldr=N)
    ref = o.float() @ wt.float().t() + bias + res
    assert got.dtype == torch.float32 and (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_pooling_tokens_upsample_gap_transposes():
    from mmfn_amd import ops
    B, S, C, T = 2, 32, 64, 192
    feats = [_r(B, S, S, C, seed=i) for i in range(3)]
    pos, vw, vb, vel = torch.randn(T, C, device=DEV), torch.randn(C, device=DEV), torch.randn(C, device=DEV), torch.rand(B, device=DEV)
    rng = torch.tensor([5, 1], dtype=torch.int64, device=DEV)
    t32 = ops.tokens_fwd([f.float() for f in feats], pos, vw, vb, vel, torch.empty(B, T, C, device=DEV), 0.1, rng, 3)
    t16 = ops.tokens_fwd(feats, pos, vw, vb, vel, torch.empty(B, T, C, dtype=BF, device=DEV), 0.1, rng, 3)
    _same(t16, t32)
    tok = _r(B, T, C, seed=9)
    for m in range(3):
        u32 = ops.upsample_add_fwd(feats[m].float(), tok.float(), torch.empty(B, S, S, C, device=DEV), m)
        u16 = ops.upsample_add_fwd(feats[m], tok, torch.empty(B, S, S, C, dtype=BF, device=DEV), m)
        _same(u16, u32)
        g32, g16 = torch.zeros(B, T, C, device=DEV), torch.zeros(B, T, C, dtype=BF, device=DEV)
        ops.upsample_adj(feats[m].float(), g32, m)
        ops.upsample_adj(feats[m], g16, m)
        _same(g16[:, m * 64:(m + 1) * 64], g32[:, m * 64:(m + 1) * 64])
        d32 = ops.pool_bcast_add(feats[m].float(), tok.float(), torch.empty(B, S, S, C, device=DEV), m)
        d16 = ops.pool_bcast_add(feats[m], tok, torch.empty(B, S, S, C, dtype=BF, device=DEV), m)
        _same(d16, d32)
    # token backward: in-place dropout mask + fp32 parameter gradients
    gt32 = tok.float().clone()
    gt16 = tok.clone()
    outs = []
    for gt in (gt32, gt16):
        dpos, dvw, dvb = torch.empty(T, C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        ops.tokens_bwd(gt, vel, dpos, dvw, dvb, 0.1, rng, 3)
        outs.append((dpos, dvw, dvb))
    _same(gt16, gt32)
    # (parameter gradients are sums of the masked values: fp32 of fp32 values vs fp32 of bf16-rounded ones)
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=2e-2, atol=2e-2 * float(a.abs().max()))
    # global average pool + branch sum, and its adjoint
    small = [_r(B, 8, 8, 512, seed=20 + i) for i in range(3)]
    p32 = ops.gap_sum_fwd([f.float() for f in small], torch.empty(B, 512, device=DEV))
    p16 = ops.gap_sum_fwd(small, torch.empty(B, 512, device=DEV))
    assert torch.equal(p32, p16)
    gq = torch.randn(B, 512, device=DEV)
    o32 = [torch.empty(B, 8, 8, 512, device=DEV) for _ in range(3)]
    o16 = [torch.empty(B, 8, 8, 512, dtype=BF, device=DEV) for _ in range(3)]
    ops.gap_sum_bwd(gq, o32)
    ops.gap_sum_bwd(gq, o16)
    _same(o16[1], o32[1])
    # max pool with saved argmax
    x = _r(B, 64, 64, 64, seed=31)
    y32, i32 = torch.empty(B, 32, 32, 64, device=DEV), torch.empty(B, 32, 32, 64, dtype=torch.uint8, device=DEV)
    y16, i16 = torch.empty(B, 32, 32, 64, dtype=BF, device=DEV), torch.empty(B, 32, 32, 64, dtype=torch.uint8, device=DEV)
    ops.maxpool_fwd(x.float(), y32, i32)
    ops.maxpool_fwd(x, y16, i16)
    _same(y16, y32)
    assert torch.equal(i16, i32)
    gy = _r(B, 32, 32, 64, seed=32)
    _same(ops.maxpool_bwd(gy, i16, torch.empty(B, 64, 64, 64, dtype=BF, device=DEV)),
          ops.maxpool_bwd(gy.float(), i32, torch.empty(B, 64, 64, 64, device=DEV)))
    # transposes across the precision boundary
    a = torch.randn(B, 64, 4096, device=DEV)
    t = ops.transpose(a, torch.empty(B, 4096, 64, dtype=BF, device=DEV), B, 64, 4096)
    _same(t, a.transpose(1, 2).contiguous())
    back = ops.transpose(t, torch.empty(B, 64, 4096, device=DEV), B, 4096, 64)
    assert torch.equal(back, t.float().transpose(1, 2).contiguous())
    cs32, cs16 = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    ops.colsum(t.view(-1, 64).float(), cs32)
    ops.colsum(t.view(-1, 64), cs16)
    assert torch.equal(cs32, cs16)


def _attn_ref(qkv, C, B, T, NH, hs, mask=None, p=0.0):
    """fp32 reference on the same bf16-rounded inputs; mask [B,NH,T,T] in {0,1} (dropout keep), packed columns [k | q | v]."""
    x = qkv.float().view(B, T, 3, NH, hs)
    k, q, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))          # [B, NH, T, hs]
    s = (q @ k.transpose(-1, -2)) * hs ** -0.5
    pr = torch.softmax(s, dim=-1)
    lse = torch.logsumexp(s, dim=-1)
    if mask is not None:
        pr = pr * mask / (1.0 - p)
    o = (pr @ v).permute(0, 2, 1, 3).reshape(B * T, C)
    return o, lse


@pytest.mark.parametrize("T", [64, 192, 256])
@pytest.mark.parametrize("hs", [16, 32, 64, 128])
def test_attention_bf16_mfma(hs, T):
    """attention16.hip (bf16 MFMA, fp32 softmax) forward and backward against torch on the same bf16-rounded q, k, v, dO;
    with dropout the keep mask is read back from the kernel itself (uniform scores, one-hot values) and fed to the reference."""
    from mmfn_amd import ops
    B, NH = 3, 4
    C = NH * hs
    rng = torch.tensor([9, 2], dtype=torch.int64, device=DEV)

    def run(qkv, dO, p):
        o, lse = torch.empty(B * T, C, dtype=BF, device=DEV), torch.empty(B, NH, T, device=DEV)
        ops.attention_fwd(qkv[:, C:2 * C], qkv, qkv[:, 2 * C:], 3 * C, o, C, lse, B, T, NH, hs, hs ** -0.5, drop_p=p, rng_state=rng, rng_stream=4)
        dqkv = torch.zeros(B * T, 3 * C, dtype=BF, device=DEV)
        delta = torch.empty(B, NH, T, device=DEV)
        if dO is not None:
            ops.attention_bwd(qkv[:, C:2 * C], qkv, qkv[:, 2 * C:], 3 * C, o, dO, C, lse, delta, dqkv[:, C:2 * C], dqkv, dqkv[:, 2 * C:], 3 * C,
                              B, T, NH, hs, hs ** -0.5, drop_p=p, rng_state=rng, rng_stream=4)
        return o, lse, dqkv

    for p in (0.0, 0.1):
        mask = None
        if p > 0.0:   # keep mask of this (rng state, stream): q = k = 0 -> uniform probabilities, V = one-hot key indicator
            mask = torch.zeros(B, NH, T, T, device=DEV)
            for k0 in range(0, T, hs):
                probe = torch.zeros(B, T, 3, NH, hs, device=DEV)
                n = min(hs, T - k0)
                for h_ in range(NH):
                    probe[:, k0:k0 + n, 2, h_, :n] = torch.eye(n, device=DEV)
                o, _, _ = run(probe.view(B * T, 3 * C).to(BF), None, p)
                got = o.float().view(B, T, NH, hs).permute(0, 2, 1, 3)[..., :n] * T * (1.0 - p)      # [B, NH, query, key k0..]
                mask[..., k0:k0 + n] = (got > 0.5).float()
            keep = float(mask.mean())
            assert abs(keep - (1.0 - p)) < 0.02, keep
        qkv = _r(B * T, 3 * C, seed=1, scale=0.7)
        dO = _r(B * T, C, seed=2)
        o, lse, dqkv = run(qkv, dO, p)
        x = qkv.float().requires_grad_(True)
        ref_o, ref_lse = _attn_ref(x, C, B, T, NH, hs, mask, p)
        ref_o.backward(dO.float())
        assert torch.allclose(lse, ref_lse, rtol=0, atol=2e-3)
        assert float((o.float() - ref_o).abs().max()) <= 2e-2 * float(ref_o.abs().max())
        err = (dqkv.float() - x.grad).abs().max(0).values.view(3, C).max(1).values
        scale = x.grad.abs().max(0).values.view(3, C).max(1).values
        assert bool((err <= 3e-2 * scale).all()), (p, err.tolist(), scale.tolist())


def test_weight_shadows():
    from mmfn_amd import ops
    w = torch.randn(96, 9, 160, device=DEV)
    lin = torch.randn(200, 72, device=DEV)
    d1, d2 = torch.zeros(w.numel(), dtype=BF, device=DEV), torch.zeros(lin.numel(), dtype=BF, device=DEV)
    table = ops.make_shadow_table([(w, d1), (lin, d2)], torch.device(DEV))
    ops.shadow_transpose(*table)
    assert torch.equal(d1.view(160, 9, 96), w.permute(2, 1, 0).contiguous().to(BF))
    assert torch.equal(d2.view(72, 200), lin.t().contiguous().to(BF))
    flat = torch.randn(1 << 16, device=DEV)
    assert torch.equal(ops.cast_to_bf16(flat, torch.empty(1 << 16, dtype=BF, device=DEV)), flat.to(BF))
