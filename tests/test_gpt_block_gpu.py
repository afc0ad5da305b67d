"""Fused GPT-block kernels (mmfn_gpt_block_*; model_vec.py:112-133) against a plain PyTorch fp64 evaluation of the same block and
against the separate HIP kernels they replace (same counter-RNG dropout masks)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _params(C, g, dev):
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    k = 1.0 / math.sqrt(C)
    return dict(ln1_w=1 + 0.1 * r(C), ln1_b=0.1 * r(C), wqkv=r(3 * C, C, sc=k), bqkv=0.1 * r(3 * C), wproj=r(C, C, sc=k),
                bproj=0.1 * r(C), ln2_w=1 + 0.1 * r(C), ln2_b=0.1 * r(C), w1=r(4 * C, C, sc=k), b1=0.1 * r(4 * C),
                w2=r(C, 4 * C, sc=0.5 * k), b2=0.1 * r(C))


def _ln(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rs = 1.0 / torch.sqrt(var + eps)
    return (x - mu) * rs * w + b, mu.squeeze(-1), rs.squeeze(-1)


def _ref_block(p, x, B, T, C, NH):
    """fp64 forward of one block without dropout; returns every saved tensor."""
    p = {k: v.double() for k, v in p.items()}
    x = x.double()
    HS = C // NH
    a, mu1, rs1 = _ln(x, p["ln1_w"], p["ln1_b"])
    qkv = a @ p["wqkv"].t() + p["bqkv"]
    k, q, v = (qkv[:, i * C:(i + 1) * C].view(B, T, NH, HS).transpose(1, 2) for i in range(3))
    att = (q @ k.transpose(-1, -2)) / math.sqrt(HS)
    lse = torch.logsumexp(att, -1)
    o = (torch.softmax(att, -1) @ v).transpose(1, 2).reshape(B * T, C)
    x1 = x + o @ p["wproj"].t() + p["bproj"]
    a2, mu2, rs2 = _ln(x1, p["ln2_w"], p["ln2_b"])
    h = torch.relu(a2 @ p["w1"].t() + p["b1"])
    x2 = x1 + h @ p["w2"].t() + p["b2"]
    return dict(a=a, mu1=mu1, rs1=rs1, qkv=qkv, o=o, lse=lse, x1=x1, a2=a2, mu2=mu2, rs2=rs2, h=h, x2=x2)


def _bufs(B, T, C, NH, dev):
    M = B * T
    e = lambda *s: torch.full(s, float("nan"), device=dev)
    return dict(a=e(M, C), mu1=e(M), rs1=e(M), qkv=e(M, 3 * C), o=e(M, C), lse=e(B, NH, T), x1=e(M, C), a2=e(M, C), mu2=e(M),
                rs2=e(M), h=e(M, 4 * C), x2=e(M, C))


def _close(got, ref, tol, name):
    ref = ref.to(got.device)
    err = (got.double() - ref.double()).abs().max().item()
    scale = ref.double().abs().max().item() + 1e-12
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e" % (name, err, scale)


@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("B", [1, 3])
def test_fused_block_forward_matches_fp64(C, B):
    from mmfn_amd import ops
    dev = _dev()
    T, NH = 192, 4
    g = torch.Generator().manual_seed(10 * C + B)
    p = _params(C, g, dev)
    x = torch.randn(B * T, C, generator=g).to(dev)
    out = _bufs(B, T, C, NH, dev)
    d = ops.gpt_block_desc(B, T, C, NH, x=x, **p, **out)
    ops.gpt_block_attn_fwd(d)
    ops.gpt_block_mlp_fwd(d)
    torch.cuda.synchronize()
    ref = _ref_block(p, x, B, T, C, NH)
    for name in ("a", "mu1", "rs1", "qkv", "lse", "o", "x1", "a2", "mu2", "rs2", "h", "x2"):
        _close(out[name], ref[name], 2e-5, name)


def _unfused_forward(ops, p, x, B, T, C, NH, out, pa, pr, rng, sb):
    """The separate kernels the engine ran up to round 5 (GPT.fwd), writing into `out`."""
    HS = C // NH
    ops.layernorm_fwd(x, p["ln1_w"], p["ln1_b"], out["a"], out["mu1"], out["rs1"])
    ops.linear_fwd(out["a"], p["wqkv"], p["bqkv"], out=out["qkv"])
    qkv = out["qkv"]
    ops.attention_fwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, out["o"], C, out["lse"], B, T, NH, HS, 1.0 / math.sqrt(HS), drop_p=pa,
                      rng_state=rng, rng_stream=sb)
    ops.linear_fwd(out["o"], p["wproj"], p["bproj"], out=out["x1"], res=x, ldr=C, drop_p=pr, rng_state=rng, rng_stream=sb + 1)
    ops.layernorm_fwd(out["x1"], p["ln2_w"], p["ln2_b"], out["a2"], out["mu2"], out["rs2"])
    ops.linear_fwd(out["a2"], p["w1"], p["b1"], out=out["h"], relu=True)
    ops.linear_fwd(out["h"], p["w2"], p["b2"], out=out["x2"], res=out["x1"], ldr=C, drop_p=pr, rng_state=rng, rng_stream=sb + 2)


@pytest.mark.parametrize("C", [64, 128])
def test_fused_block_forward_with_dropout_equals_the_separate_kernels(C):
    """Same counter-RNG masks (attention: stream sb, proj: sb + 1, mlp.2: sb + 2): fused and unfused forwards agree to rounding."""
    from mmfn_amd import ops
    dev = _dev()
    B, T, NH = 4, 192, 4
    g = torch.Generator().manual_seed(C)
    p = _params(C, g, dev)
    x = torch.randn(B * T, C, generator=g).to(dev)
    rng = torch.tensor([11, 3], dtype=torch.int64, device=dev)
    fused, plain = _bufs(B, T, C, NH, dev), _bufs(B, T, C, NH, dev)
    d = ops.gpt_block_desc(B, T, C, NH, attn_pdrop=0.1, resid_pdrop=0.1, rng_state=rng, rng_stream=40, x=x, **p, **fused)
    ops.gpt_block_attn_fwd(d)
    ops.gpt_block_mlp_fwd(d)
    _unfused_forward(ops, p, x, B, T, C, NH, plain, 0.1, 0.1, rng, 40)
    torch.cuda.synchronize()
    for name in fused:
        _close(fused[name], plain[name], 2e-5, name)
    # the masks really dropped something
    assert (fused["x1"] - x - (fused["o"] @ p["wproj"].t() + p["bproj"])).abs().max().item() > 1e-3


@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("pr", [0.0, 0.1])
def test_fused_block_backward_rows_equal_the_separate_kernels(C, pr):
    """mmfn_gpt_block_bwd_rows_f32 (upper + lower in one launch, and each alone) against the chain of separate kernels of GPT.bwd:
    mlp.2 dgrad with the ReLU mask, mlp.0 dgrad, ln2 backward (+ dropped copy + partial rows), proj dgrad; qkv dgrad, ln1 backward."""
    from mmfn_amd import ops
    dev = _dev()
    B, T, NH = 2, 192, 4
    M = B * T
    g = torch.Generator().manual_seed(7 * C)
    pu, pl = _params(C, g, dev), _params(C, g, dev)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    rng = torch.tensor([5, 9], dtype=torch.int64, device=dev)
    sb_up, sb_lo = 50, 47
    # forward state of both blocks (any consistent values do: the backward kernels only read them)
    x_lo = r(M, C)
    f_lo = _bufs(B, T, C, NH, dev)
    _unfused_forward(ops, pl, x_lo, B, T, C, NH, f_lo, 0.0, pr, rng, sb_lo)
    x_up = f_lo["x2"].clone()
    f_up = _bufs(B, T, C, NH, dev)
    _unfused_forward(ops, pu, x_up, B, T, C, NH, f_up, 0.0, pr, rng, sb_up)
    dqkv, g1_up = r(M, 3 * C), r(M, C)
    drop = pr > 0
    nrow = M // ops.GPT_ROWS
    e = lambda *s: torch.full(s, float("nan"), device=dev)

    def run_fused(split):
        o = dict(g_below=e(M, C), gd_below=e(M, C) if drop else None, part_ln1=e(nrow, 3, C), gh=e(M, 4 * C), g1=e(M, C),
                 gd2=e(M, C) if drop else None, go=e(M, C), part_ln2=e(nrow, 3, C))
        up = ops.gpt_block_desc(B, T, C, NH, resid_pdrop=pr, rng_state=rng, rng_stream=sb_up, rng_stream_below=sb_lo, below_colsum=True,
                                x=x_up, mu1=f_up["mu1"], rs1=f_up["rs1"], dqkv=dqkv, g1=g1_up, g_below=o["g_below"],
                                gd_below=o["gd_below"], part_ln1=o["part_ln1"], **pu)
        lo = ops.gpt_block_desc(B, T, C, NH, resid_pdrop=pr, rng_state=rng, rng_stream=sb_lo, x1=f_lo["x1"], mu2=f_lo["mu2"],
                                rs2=f_lo["rs2"], h=f_lo["h"], g=o["g_below"], gd=o["gd_below"], gh=o["gh"], g1=o["g1"],
                                gd2=o["gd2"], go=o["go"], part_ln2=o["part_ln2"], **pl)
        if split:
            ops.gpt_block_bwd_rows(up, None)
            ops.gpt_block_bwd_rows(None, lo)
        else:
            ops.gpt_block_bwd_rows(up, lo)
        gw1, gb1, cs1, gw2, gb2, cs2 = e(C), e(C), e(C), e(C), e(C), e(C)
        ops.layernorm_bwd_finalize(o["part_ln1"], nrow, C, gw1, gb1, cs1)
        ops.layernorm_bwd_finalize(o["part_ln2"], nrow, C, gw2, gb2, cs2)
        o.update(ln1_gw=gw1, ln1_gb=gb1, cs_below=cs1, ln2_gw=gw2, ln2_gb=gb2, cs_proj=cs2)
        return o

    # the separate kernels
    ref = {}
    ga = ops.linear_dx(dqkv, pu["wqkv"])
    ref["g_below"], ref["gd_below"] = e(M, C), e(M, C) if drop else None
    ref["ln1_gw"], ref["ln1_gb"], ref["cs_below"] = e(C), e(C), e(C)
    ops.layernorm_bwd(ga, x_up, pu["ln1_w"], pu["ln1_b"], f_up["mu1"], f_up["rs1"], ref["g_below"], ref["ln1_gw"], ref["ln1_gb"], 0,
                      dres=g1_up, dx_dropped=ref["gd_below"], drop_p=pr, rng_state=rng if drop else None, rng_stream=sb_lo + 2,
                      dx_colsum=ref["cs_below"])
    gp = ref["gd_below"] if drop else ref["g_below"]
    ref["gh"] = e(M, 4 * C)
    ops.linear_dx(gp, pl["w2"], out=ref["gh"], aux=f_lo["h"], ldaux=4 * C)
    ga2 = ops.linear_dx(ref["gh"], pl["w1"])
    ref["g1"], ref["gd2"] = e(M, C), e(M, C) if drop else None
    ref["ln2_gw"], ref["ln2_gb"], ref["cs_proj"] = e(C), e(C), e(C)
    ops.layernorm_bwd(ga2, f_lo["x1"], pl["ln2_w"], pl["ln2_b"], f_lo["mu2"], f_lo["rs2"], ref["g1"], ref["ln2_gw"], ref["ln2_gb"], 0,
                      dres=ref["g_below"], dx_dropped=ref["gd2"], drop_p=pr, rng_state=rng if drop else None, rng_stream=sb_lo + 1,
                      dx_colsum=ref["cs_proj"])
    ref["go"] = ops.linear_dx(ref["gd2"] if drop else ref["g1"], pl["wproj"])
    torch.cuda.synchronize()
    for split in (False, True):
        got = run_fused(split)
        torch.cuda.synchronize()
        for name, want in ref.items():
            if want is not None:
                _close(got[name], want, 3e-5, "%s (split=%s)" % (name, split))


# ------------------------------------------------------------------------------------------------- bf16 training mode
def _close16(got, ref, name, ulps=2.0):
    """bf16 tensors: within `ulps` bf16 ulps of the reference's magnitude scale (the fused and the separate kernels round the same
    fp32 accumulations, in another summation order); fp32 tensors: 1e-2 of the scale (they inherit bf16 operands)."""
    g, r = got.double(), ref.to(got.device).double()
    scale = r.abs().max().item() + 1e-12
    tol = (ulps * 2.0 ** -8 if got.dtype == torch.bfloat16 else 1e-2) * scale
    err = (g - r).abs().max().item()
    assert err <= tol, "%s: max err %.3e > %.3e (scale %.3e)" % (name, err, tol, scale)


def _shadows(p):
    bf = torch.bfloat16
    s = {k: p[k] for k in ("ln1_w", "ln1_b", "bqkv", "bproj", "ln2_w", "ln2_b", "b1", "b2")}
    fwd = dict(s, **{k: p[k].to(bf).contiguous() for k in ("wqkv", "wproj", "w1", "w2")})
    bwd = dict(s, **{k: p[k].t().contiguous().to(bf) for k in ("wqkv", "wproj", "w1", "w2")})
    return fwd, bwd


@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("pr", [0.0, 0.1])
def test_fused_mlp_forward_bf16_equals_the_separate_bf16_kernels(C, pr):
    from mmfn_amd import ops
    dev = _dev()
    bf = torch.bfloat16
    B, T, NH = 3, 192, 4
    M = B * T
    g = torch.Generator().manual_seed(3 * C)
    p = _params(C, g, dev)
    fw, _ = _shadows(p)
    x = torch.randn(M, C, generator=g).to(dev)
    o = torch.randn(M, C, generator=g).to(dev).to(bf)
    rng = torch.tensor([21, 4], dtype=torch.int64, device=dev)
    e32 = lambda *s: torch.full(s, float("nan"), device=dev)
    e16 = lambda *s: torch.full(s, float("nan"), device=dev, dtype=bf)
    got = dict(x1=e32(M, C), a2=e16(M, C), mu2=e32(M), rs2=e32(M), h=e16(M, 4 * C), x2=e32(M, C))
    d = ops.gpt_block_desc(B, T, C, NH, resid_pdrop=pr, rng_state=rng, rng_stream=60, x=x, o=o, **fw, **got)
    assert d.bf16
    ops.gpt_block_mlp_fwd(d)
    ref = dict(x1=e32(M, C), a2=e16(M, C), mu2=e32(M), rs2=e32(M), h=e16(M, 4 * C), x2=e32(M, C))
    ops.linear_fwd(o, fw["wproj"], p["bproj"], out=ref["x1"], res=x, ldr=C, drop_p=pr, rng_state=rng, rng_stream=61)
    ops.layernorm_fwd(ref["x1"], p["ln2_w"], p["ln2_b"], ref["a2"], ref["mu2"], ref["rs2"])
    ops.linear_fwd(ref["a2"], fw["w1"], p["b1"], out=ref["h"], relu=True)
    ops.linear_fwd(ref["h"], fw["w2"], p["b2"], out=ref["x2"], res=ref["x1"], ldr=C, drop_p=pr, rng_state=rng, rng_stream=62)
    torch.cuda.synchronize()
    for name in got:
        _close16(got[name], ref[name], name)


@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("pr", [0.0, 0.1])
def test_fused_backward_rows_bf16_equal_the_separate_bf16_kernels(C, pr):
    from mmfn_amd import ops
    dev = _dev()
    bf = torch.bfloat16
    B, T, NH = 2, 192, 4
    M = B * T
    g = torch.Generator().manual_seed(11 * C)
    pu, pl = _params(C, g, dev), _params(C, g, dev)
    _, bu = _shadows(pu)
    _, bl = _shadows(pl)
    r32 = lambda *s: torch.randn(*s, generator=g).to(dev)
    rng = torch.tensor([5, 9], dtype=torch.int64, device=dev)
    sb_up, sb_lo = 50, 47
    x_up, x1_lo = r32(M, C), r32(M, C)
    mu1, rs1 = x_up.mean(1), 1.0 / torch.sqrt(x_up.var(1, unbiased=False) + 1e-5)
    mu2, rs2 = x1_lo.mean(1), 1.0 / torch.sqrt(x1_lo.var(1, unbiased=False) + 1e-5)
    h = torch.relu(r32(M, 4 * C)).to(bf)
    dqkv, g1_up = r32(M, 3 * C).to(bf), r32(M, C)
    nrow = M // ops.GPT_ROWS
    e32 = lambda *s: torch.full(s, float("nan"), device=dev)
    e16 = lambda *s: torch.full(s, float("nan"), device=dev, dtype=bf)
    o = dict(g_below=e32(M, C), gd_below=e16(M, C), part_ln1=e32(nrow, 3, C), gh=e16(M, 4 * C), g1=e32(M, C), gd2=e16(M, C),
             go=e16(M, C), part_ln2=e32(nrow, 3, C))
    up = ops.gpt_block_desc(B, T, C, NH, resid_pdrop=pr, rng_state=rng, rng_stream=sb_up, rng_stream_below=sb_lo, below_colsum=True,
                            x=x_up, mu1=mu1, rs1=rs1, dqkv=dqkv, g1=g1_up, g_below=o["g_below"], gd_below=o["gd_below"],
                            part_ln1=o["part_ln1"], **bu)
    lo = ops.gpt_block_desc(B, T, C, NH, resid_pdrop=pr, rng_state=rng, rng_stream=sb_lo, x1=x1_lo, mu2=mu2, rs2=rs2, h=h,
                            g=o["g_below"], gd=o["gd_below"], gh=o["gh"], g1=o["g1"], gd2=o["gd2"], go=o["go"], part_ln2=o["part_ln2"],
                            **bl)
    assert up.bf16 and lo.bf16
    ops.gpt_block_bwd_rows(up, lo)
    got = dict(o)
    for tag, part in (("1", o["part_ln1"]), ("2", o["part_ln2"])):
        gw, gb, cs = e32(C), e32(C), e32(C)
        ops.layernorm_bwd_finalize(part, nrow, C, gw, gb, cs)
        got.update({"ln%s_gw" % tag: gw, "ln%s_gb" % tag: gb, "cs%s" % tag: cs})
    # the separate kernels of GPT.bwd in the bf16 mode
    ref = {}
    ga = ops.linear_dx(dqkv, bu["wqkv"], out=e16(M, C))
    ref["g_below"], ref["gd_below"] = e32(M, C), e16(M, C)
    ref["ln1_gw"], ref["ln1_gb"], ref["cs1"] = e32(C), e32(C), e32(C)
    ops.layernorm_bwd(ga, x_up, pu["ln1_w"], pu["ln1_b"], mu1, rs1, ref["g_below"], ref["ln1_gw"], ref["ln1_gb"], 0, dres=g1_up,
                      dx_dropped=ref["gd_below"], drop_p=pr, rng_state=rng, rng_stream=sb_lo + 2, dx_colsum=ref["cs1"])
    ref["gh"] = e16(M, 4 * C)
    ops.linear_dx(ref["gd_below"], bl["w2"], out=ref["gh"], aux=h, ldaux=4 * C)
    ga2 = ops.linear_dx(ref["gh"], bl["w1"], out=e16(M, C))
    ref["g1"], ref["gd2"] = e32(M, C), e16(M, C)
    ref["ln2_gw"], ref["ln2_gb"], ref["cs2"] = e32(C), e32(C), e32(C)
    ops.layernorm_bwd(ga2, x1_lo, pl["ln2_w"], pl["ln2_b"], mu2, rs2, ref["g1"], ref["ln2_gw"], ref["ln2_gb"], 0, dres=ref["g_below"],
                      dx_dropped=ref["gd2"], drop_p=pr, rng_state=rng, rng_stream=sb_lo + 1, dx_colsum=ref["cs2"])
    ref["go"] = ops.linear_dx(ref["gd2"], bl["wproj"], out=e16(M, C))
    torch.cuda.synchronize()
    for name, want in ref.items():
        # (the separate path rounds ga / ga2 to bf16 before the LayerNorm backward, the fused one keeps them fp32: a few ulps)
        _close16(got[name], want, name, ulps=4.0)
