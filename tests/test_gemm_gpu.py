"""GPU parity of the fp32 MFMA GEMM / implicit-conv kernel vs plain torch fp32 on CPU."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda:0")


def _close(got, ref, tol=2e-4):
    got = got.detach().cpu().double()
    ref = ref.double()
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, "max err %g vs scale %g" % (err, scale)


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (128, 128, 64), (100, 70, 36), (6144, 192, 64),
                                   (333, 257, 129), (32, 256, 512), (2048, 512, 4608), (37, 64, 7), (8, 3, 1)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7])
def test_linear_forms(M, N, K, tile):
    from mmfn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xd, wd, bd, dyd = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)
    _close(ops.linear_fwd(xd, wd, bd, tile=tile), x @ w.t() + b)
    _close(ops.linear_fwd(xd, wd, bd, relu=True, tile=tile), torch.relu(x @ w.t() + b))
    _close(ops.linear_dx(dyd, wd, tile=tile), dy @ w)
    _close(ops.linear_dw(dyd, xd, tile=tile), dy.t() @ x)
    # forced split-K must agree
    _close(ops.linear_dw(dyd, xd, tile=tile, splitk=3), dy.t() @ x)
    _close(ops.linear_fwd(xd, wd, bd, tile=tile, splitk=2), x @ w.t() + b)


def test_epilogue_flags():
    from mmfn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    M, N, K = 192, 128, 64
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    r, aux = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    c0 = torch.randn(M, N, generator=g)
    xd, wd, bd, rd, auxd = (t.to(dev) for t in (x, w, b, r, aux))
    _close(ops.linear_fwd(xd, wd, bd, res=rd, ldr=N), x @ w.t() + b + r)
    _close(ops.linear_fwd(xd, wd, bd, gelu=True), F.gelu(x @ w.t() + b))
    _close(ops.linear_fwd(xd, wd, None, aux=auxd, ldaux=N), (x @ w.t()) * (aux > 0))
    out = c0.to(dev).clone()
    _close(ops.linear_fwd(xd, wd, bd, out=out, accum=True), x @ w.t() + b + c0)
    # dropout: kept entries are scaled by 1/(1-p), drop rate ~ p, same mask when re-run
    state = torch.tensor([1234, 7], dtype=torch.int64, device=dev)
    y1 = ops.linear_fwd(xd, wd, bd, drop_p=0.25, rng_state=state, rng_stream=3).cpu()
    y2 = ops.linear_fwd(xd, wd, bd, drop_p=0.25, rng_state=state, rng_stream=3).cpu()
    assert torch.equal(y1, y2)
    ref = x @ w.t() + b
    kept = y1 != 0
    assert 0.70 < kept.float().mean().item() < 0.80
    _close(y1[kept], (ref / 0.75)[kept])
    y3 = ops.linear_fwd(xd, wd, bd, drop_p=0.25, rng_state=state, rng_stream=4).cpu()
    assert not torch.equal(y1 != 0, y3 != 0)


@pytest.mark.parametrize("M,C,N,relu", [(384, 64, 192, False), (6144, 512, 1536, False), (6144, 512, 2048, True), (1024, 256, 768, False),
                                          (768, 128, 512, True)])
def test_layernorm_folded_into_the_gemm(M, C, N, relu):
    """MMFN_EPI_LN_FOLD: LN(x) W^T + b as ONE launch - the GEMM runs on the raw rows against W . diag(gamma) and applies
    rstd * (acc - mean * c1) + c2 in its epilogue, the row statistics accumulated from the A fragments (model_vec.py:117-118 ->
    82-98 ln1 -> key/query/value, :119-121 ln2 -> mlp.0).  Against layernorm_fwd + linear_fwd of the same operands, every
    tile shape that divides (M, N); the statistics it writes against the LayerNorm kernel's."""
    from mmfn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + C + N)
    x = (torch.randn(M, C, generator=g) * 1.7 + 0.6).to(dev)          # residual-stream-like: mean of the order of the spread
    w = (torch.randn(N, C, generator=g) * C ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    a, mu0, rs0 = torch.empty(M, C, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(x, gamma, beta, a, mu0, rs0)
    ref = ops.linear_fwd(a, w, b, relu=relu)
    wf, c1, c2 = torch.empty_like(w), torch.empty(N, device=dev), torch.empty(N, device=dev)
    ops.ln_fold_weights(*ops.make_ln_fold_table([(w, gamma, beta, b, wf, c1, c2)], dev))
    assert (wf - w * gamma).abs().max().item() == 0.0
    assert (c1 - (w * gamma).double().sum(1).float()).abs().max().item() <= 1e-6 * max(1.0, c1.abs().max().item())
    assert (c2 - ((w.double() * beta.double()).sum(1) + b.double()).float()).abs().max().item() <= 1e-6 * max(1.0, c2.abs().max().item())
    scale = ref.abs().max().item()
    tiles = [0] + [t for t, (bm, bn) in ((1, (128, 128)), (2, (64, 64)), (3, (128, 64)), (4, (64, 128))) if M % bm == 0 and N % bn == 0]
    for tile in tiles:
        mu, rs = torch.full((M,), float("nan"), device=dev), torch.full((M,), float("nan"), device=dev)
        out = torch.full((M, N), float("nan"), device=dev)
        ops.linear_fwd(x, wf, c2, out=out, relu=relu, tile=tile, ln_fold=(c1, mu, rs, 1e-5))
        err = (out - ref).abs().max().item()
        assert err <= 3e-5 * scale, (tile, err, scale)
        assert (mu - mu0).abs().max().item() <= 1e-6 * max(1.0, mu0.abs().max().item())
        assert ((rs - rs0) / rs0).abs().max().item() <= 2e-5, ((rs - rs0) / rs0).abs().max().item()
    # a tile that does not divide N: the wrapper falls back to 64 x 64 (the C entry itself refuses)
    if N % 128:
        out = ops.gemm(x, wf, torch.empty(M, N, device=dev), M, N, C, C, C, N, bias=c2, relu=relu, tile=1, ln_fold=(c1, None, None, 1e-5))
        assert (out - ref).abs().max().item() <= 3e-5 * scale


@pytest.mark.parametrize("relu", [False, True])
def test_nan_propagates_alike_through_interior_and_edge_tiles(relu):
    """A NaN accumulator must leave the GEMM the same way from an interior tile (flag-hoisted epilogue) and from an edge tile
    (general per-element epilogue): NaN without ReLU - never -inf, which would mask a diverged run in one part of the matrix
    only - and fmaxf's 0 with it, as torch.relu(nan) is NOT what the reference gives but both tile kinds agree."""
    from mmfn_amd import ops
    dev = _dev()
    M, N, K = 192 + 40, 128 + 24, 64          # 64x64 tiles: interior tiles and a ragged right / bottom edge
    g = torch.Generator().manual_seed(17)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    x[5, 3] = float("nan")      # row of an interior tile
    x[M - 2, 7] = float("nan")  # row of an edge tile
    for tile in (0, 1, 4):
        y = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev), relu=relu, tile=tile).cpu()
        for row in (5, M - 2):
            if relu:
                assert bool((y[row] == 0).all()), (tile, row)
            else:
                assert bool(torch.isnan(y[row]).all()), (tile, row, y[row][:4])
        ok = torch.ones(M, dtype=torch.bool); ok[5] = ok[M - 2] = False
        ref = x[ok] @ w.t() + b
        _close(y[ok], torch.relu(ref) if relu else ref)


CONVS = [  # B, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 64, 64, 3, 1, 1), (2, 16, 16, 64, 128, 3, 2, 1), (2, 16, 16, 64, 128, 1, 2, 0),
    (1, 8, 8, 256, 512, 3, 2, 1), (3, 9, 11, 16, 32, 3, 1, 1), (2, 32, 32, 3, 64, 7, 2, 3),
    (2, 32, 32, 2, 64, 7, 2, 3), (2, 8, 8, 512, 512, 3, 1, 1),
]


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7])
def test_conv_forms(cfg, tile):
    from mmfn_amd import ops
    dev = _dev()
    B, H, W, Cin, Cout, k, s, p = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    y_ref = F.conv2d(x, w, stride=s, padding=p)
    dy = torch.randn(y_ref.shape, generator=g)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous().to(dev)
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    y = ops.conv2d_fwd(x_nhwc, w_ohwi, s, p, tile=tile)
    _close(y.permute(0, 3, 1, 2), y_ref)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, stride=s, padding=p).backward(dy)
    if Cin % 4 == 0:
        dx = ops.conv2d_dgrad(dy_nhwc, w_ohwi, tuple(x_nhwc.shape), s, p, tile=tile)
        _close(dx.permute(0, 3, 1, 2), xr.grad)
    dw = ops.conv2d_wgrad(dy_nhwc, x_nhwc, tuple(w_ohwi.shape), s, p, tile=tile)
    _close(dw.permute(0, 3, 1, 2), wr.grad)
    dw2 = ops.conv2d_wgrad(dy_nhwc, x_nhwc, tuple(w_ohwi.shape), s, p, tile=tile, splitk=4)
    _close(dw2.permute(0, 3, 1, 2), wr.grad)


@pytest.mark.parametrize("cfg", [(3, 16, 16, 256, 256), (2, 8, 8, 512, 512), (1, 4, 6, 256, 512)])
def test_winograd_conv_and_dgrad(cfg):
    """Winograd F(2x2,3x3) path (taken from 256 channels up) against torch and against the implicit-GEMM path."""
    from mmfn_amd import ops
    dev = _dev()
    B, H, W, Cin, Cout = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, w, padding=1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous().to(dev)
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    assert ops.winograd_ok(x_nhwc.shape, w_ohwi.shape, 1, 1, {})
    y = ops.conv2d_fwd(x_nhwc, w_ohwi, 1, 1)
    _close(y.permute(0, 3, 1, 2), y_ref.detach())
    direct = ops.conv2d_fwd(x_nhwc, w_ohwi, 1, 1, tile=1)  # explicit tile -> implicit GEMM
    assert (y - direct).abs().max().item() <= 2e-5 * direct.abs().max().item()
    for m in ((2, 4) if H % 4 == 0 and W % 4 == 0 else (2,)):  # both transform sizes explicitly
        ym = ops.conv2d_winograd(x_nhwc, w_ohwi, torch.empty_like(direct), m=m)
        err = (ym - direct).abs().max().item() / direct.abs().max().item()
        assert err <= (2e-5 if m == 4 else 5e-6), (m, err)
    res = torch.randn(B, H, W, Cin, generator=g).to(dev)
    dx = ops.conv2d_dgrad(dy_nhwc, w_ohwi, tuple(x_nhwc.shape), 1, 1, res=res, ldr=Cin)
    _close((dx - res).permute(0, 3, 1, 2), xr.grad)
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, padding=1).backward(dy)
    dw = ops.conv2d_wgrad(dy_nhwc, x_nhwc, tuple(w_ohwi.shape), 1, 1)  # Winograd domain when H, W are multiples of 4
    _close(dw.permute(0, 3, 1, 2), wr.grad)
    dw_direct = ops.conv2d_wgrad(dy_nhwc, x_nhwc, tuple(w_ohwi.shape), 1, 1, tile=2)
    assert (dw - dw_direct).abs().max().item() <= 3e-5 * dw_direct.abs().max().item()


@pytest.mark.parametrize("cfg", [(4, 32, 64, 64), (4, 16, 256, 256)])
def test_winograd_f4_rounding_error_against_fp64(cfg):
    """F(4x4,3x3) over the points 0, +-3/4, +-3/2, inf (csrc/winograd.hip): forward, data gradient (adjoint pipeline) and weight
    gradient against an fp64 convolution, with the direct implicit GEMM's error on the same operands as the yardstick.  Measured
    ~5-7x the direct form's rms error (Lavin & Gray's points 0, +-1, +-2 of rounds 1-3: ~12x; tools/experiments/winograd_points.py)."""
    from mmfn_amd import ops
    dev = _dev()
    B, HW, Cin, Cout = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.relu(torch.randn(B, Cin, HW, HW, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    dy = torch.randn(B, Cout, HW, HW, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(xr, wr, padding=1)
    y64.backward(dy.double())
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    xd, wd, dyd = nhwc(x), nhwc(w), nhwc(dy)
    rms = lambda got, ref: float((got.detach().cpu().double().permute(0, 3, 1, 2) - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    u = torch.empty(36 * Cout * Cin, device=dev)
    v = torch.empty(ops.winograd_v_numel(xd.shape), device=dev)
    y = ops.conv2d_fwd(xd, wd, 1, 1, keep_v=v, keep_u=u)
    dw, dx = torch.empty_like(wd), torch.empty_like(xd)
    ops.conv2d_bwd_winograd(dyd, xd, u, dw, dx, v=v)
    e_w = (rms(y, y64.detach()), rms(dx, xr.grad), rms(dw, wr.grad))
    e_d = (rms(ops.conv2d_fwd(xd, wd, 1, 1, tile=1), y64.detach()),
           rms(ops.gemm(dyd, wd, torch.empty_like(xd), B * HW * HW, Cin, 9 * Cout, 0, 0, Cin, ops.A_DGRAD, ops.B_DGRADW,
                        conv=ops.conv_geom(xd.shape, wd.shape, 1, 1)[0]), xr.grad),
           rms(ops.conv2d_wgrad(dyd, xd, tuple(wd.shape), 1, 1, tile=2), wr.grad))
    print("\n[F(4x4,3x3) %s] rel rms error vs fp64: winograd fwd %.2e dgrad %.2e wgrad %.2e | direct %.2e %.2e %.2e" % ((cfg,) + e_w + e_d))
    for a, b in zip(e_w, e_d):
        assert a <= 9.0 * b and a <= 2.5e-6, (e_w, e_d)


@pytest.mark.parametrize("cfg", [(3, 32, 128, 128), (2, 16, 256, 256), (5, 8, 512, 512), (2, 16, 128, 256)])
def test_winograd_adjoint_backward(cfg):
    """Weight + data gradient together in the F(4x4,3x3) domain (ops.conv2d_bwd_winograd): the data gradient is the adjoint of
    the forward pipeline (dM . U, overlap-add of B dV B^T) and reuses the transformed filter the forward kept; against torch,
    with and without the residual add, with the kept / recomputed transformed input, and bitwise repeatable."""
    from mmfn_amd import ops
    dev = _dev()
    B, HW, Cin, Cout = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, Cin, HW, HW, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, padding=1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous().to(dev)
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    assert ops.winograd_adjoint_ok(x_nhwc.shape, w_ohwi.shape, 1, 1)
    u = torch.empty(36 * Cout * Cin, device=dev)
    v = torch.empty(ops.winograd_v_numel(x_nhwc.shape), device=dev)
    y = ops.conv2d_fwd(x_nhwc, w_ohwi, 1, 1, keep_v=v, keep_u=u)
    _close(y.permute(0, 3, 1, 2), y_ref.detach())
    res = torch.randn(B, HW, HW, Cin, generator=g).to(dev)
    outs = []
    for kept_v, r in ((v, None), (None, res), (v, res)):
        dw, dx = torch.empty_like(w_ohwi), torch.empty_like(x_nhwc)
        ops.conv2d_bwd_winograd(dy_nhwc, x_nhwc, u, dw, dx, v=kept_v, res=r)
        _close(dw.permute(0, 3, 1, 2), wr.grad)
        _close((dx if r is None else dx - r).permute(0, 3, 1, 2), xr.grad)
        outs.append((dw, dx))
    assert torch.equal(outs[1][1], outs[2][1]) and torch.equal(outs[0][0], outs[2][0])
    # against the flipped-filter data gradient (the path it replaces)
    old = ops.conv2d_dgrad(dy_nhwc, w_ohwi, tuple(x_nhwc.shape), 1, 1)
    assert (outs[0][1] - old).abs().max().item() <= 3e-5 * old.abs().max().item()


@pytest.mark.parametrize("relu", [True, False])
def test_winograd_backward_with_fused_batchnorm_backward(relu):
    """conv2d_bwd_winograd(bn=...): the BatchNorm backward of g is formed inside the output-gradient transform (its reductions
    done by bn_bwd_reduce) instead of by bn_bwd's apply pass - same dx / dw / parameter gradients / masked g as the two-pass path."""
    from mmfn_amd import ops
    dev = _dev()
    B, HW, Cin, Cout = 3, 16, 128, 256
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, HW, HW, Cin, generator=gen).to(dev)
    w = (torch.randn(Cout, 3, 3, Cin, generator=gen) * 0.05).to(dev)
    u = torch.empty(36 * Cout * Cin, device=dev)
    v = torch.empty(ops.winograd_v_numel(x.shape), device=dev)
    co = ops.conv2d_fwd(x, w, 1, 1, keep_v=v, keep_u=u)
    M = B * HW * HW
    mean, var = co.view(M, Cout).mean(0), co.view(M, Cout).var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    bn_w = (torch.rand(Cout, generator=gen) + 0.5).to(dev)
    y = ((co - mean) * rstd * bn_w + 0.1)
    y = torch.relu(y) if relu else y
    g = torch.randn(B, HW, HW, Cout, generator=gen).to(dev)
    ymask = y.view(M, Cout) if relu else None
    # two passes
    dco, ge_a = torch.empty_like(co), torch.empty_like(g)
    dbw_a, dbb_a = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    ops.bn_bwd(g.view(M, Cout), ymask, co.view(M, Cout), mean, rstd, bn_w, dco.view(M, Cout), dbw_a, dbb_a, ge_out=ge_a.view(M, Cout))
    dw_a, dx_a = torch.empty_like(w), torch.empty_like(x)
    ops.conv2d_bwd_winograd(dco, x, u, dw_a, dx_a, v=v)
    # fused
    means = torch.empty(2, Cout, device=dev)
    dbw_b, dbb_b = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    ops.bn_bwd_reduce(g.view(M, Cout), ymask, co.view(M, Cout), mean, rstd, dbw_b, dbb_b, means)
    dw_b, dx_b, ge_b = torch.empty_like(w), torch.empty_like(x), torch.empty_like(g)
    ops.conv2d_bwd_winograd(dco, x, u, dw_b, dx_b, v=v, bn=(g, None if ymask is None else y, co, mean, rstd, bn_w, None, means, ge_b))
    assert torch.equal(dbw_a, dbw_b) and torch.equal(dbb_a, dbb_b) and torch.equal(ge_a, ge_b)
    for a, b in ((dx_a, dx_b), (dw_a, dw_b)):
        assert (a - b).abs().max().item() <= 2e-6 * a.abs().max().item()   # fma contraction may differ between the two kernels


@pytest.mark.parametrize("shape", [(384, 256, 512), (224, 160, 96), (6144, 512, 2048), (64, 2048, 6144)])
def test_bf16_operand_gemm_forms(shape):
    """MMFN_EPI_BF16_OPERANDS: every plain form equals the fp32 product of the bf16-rounded operands (fp32 accumulate);
    epilogues, split-K and batching behave as in the fp32 kernel."""
    from mmfn_amd import ops
    dev = _dev()
    M, N, K = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    dy = torch.randn(M, N, generator=g)
    b = torch.randn(N, generator=g)
    rnd = lambda t: t.bfloat16().float()
    xd, wd, dyd, bd = x.to(dev), w.to(dev), dy.to(dev), b.to(dev)

    def close(got, ref):
        err = (got.cpu() - ref).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err

    with ops.precision("bf16"):
        close(ops.linear_fwd(xd, wd, bd, relu=True), torch.relu(rnd(x) @ rnd(w).t() + b))       # NT
        close(ops.linear_dx(dyd, wd), rnd(dy) @ rnd(w))                                        # NN (n-contiguous weights)
        close(ops.linear_dw(dyd, xd), rnd(dy).t() @ rnd(x))                                    # TN
        close(ops.linear_dw(dyd, xd, splitk=3), rnd(dy).t() @ rnd(x))
        res = torch.randn(M, N, generator=g)
        close(ops.linear_fwd(xd, wd, bd, res=res.to(dev), ldr=N), rnd(x) @ rnd(w).t() + b + res)
    # outside the context the same call is the fp32 path again; so is a contraction length that is not a multiple of 32
    ref32 = x @ w.t() + b
    assert (ops.linear_fwd(xd, wd, bd).cpu() - ref32).abs().max().item() <= 1e-4 * ref32.abs().max().item()
    with ops.precision("bf16"):
        got = ops.linear_fwd(xd[:, :K - 8].contiguous(), wd[:, :K - 8].contiguous(), bd).cpu()
    ref = x[:, :K - 8] @ w[:, :K - 8].t() + b
    assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("cfg", [(2, 32, 64, 64, 3, 1), (2, 32, 64, 128, 3, 2), (3, 16, 128, 128, 3, 1), (2, 16, 256, 512, 1, 2),
                                 (2, 8, 512, 512, 3, 1)])
def test_bf16_direct_convolution_forms(cfg):
    """bf16 mode runs the convolutions as DIRECT implicit GEMMs on the bf16 MFMA pipe (no Winograd domain rounding): forward,
    stride-1 data gradient (flipped filter) and weight gradient equal torch's convolution of the bf16-rounded operands
    accumulated in fp32; the stride-2 data gradient stays on the fp32 kernel and equals the fp32 result."""
    from mmfn_amd import ops
    dev = _dev()
    B, H, Cin, Cout, k, st = cfg
    p = k // 2
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    rnd = lambda t: t.bfloat16().float()
    xr, wr = rnd(x).requires_grad_(True), rnd(w).requires_grad_(True)
    y_ref = F.conv2d(xr, wr, stride=st, padding=p)
    dy = torch.randn(y_ref.shape, generator=g)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous().to(dev)
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous().to(dev)

    def close(got, ref, tol=3e-5):
        err = (got.cpu() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), err

    with ops.precision("bf16"):
        assert not ops.winograd_ok(x_nhwc.shape, w_ohwi.shape, st, p, {})
        y = ops.conv2d_fwd(x_nhwc, w_ohwi, st, p)
        close(y.permute(0, 3, 1, 2), y_ref.detach())
        # weight gradient: kept in fp32 in bf16 mode (Winograd domain where that applies) - equals the unrounded result
        dw = ops.conv2d_wgrad(dy_nhwc, x_nhwc, tuple(w_ohwi.shape), st, p)
        close(dw.permute(0, 3, 1, 2), torch.nn.grad.conv2d_weight(x, w.shape, dy, stride=st, padding=p), 1e-4)
        # ... the bf16 kernel's own weight-gradient form (MMFN_BF16_WGRAD=1): dw = conv(round(x), round(dy))
        ops.BF16_WGRAD = True
        try:
            g_, _ = ops.conv_geom(x_nhwc.shape, w_ohwi.shape, st, p)
            dw16 = torch.empty_like(w_ohwi)
            ops.gemm(dy_nhwc, x_nhwc, dw16, Cout, k * k * Cin, dy_nhwc.numel() // Cout, Cout, 0, k * k * Cin, ops.A_COLMAJOR,
                     ops.B_IM2COL, conv=g_)
        finally:
            ops.BF16_WGRAD = False
        close(dw16.permute(0, 3, 1, 2), torch.nn.grad.conv2d_weight(rnd(x), w.shape, rnd(dy), stride=st, padding=p))
        # data gradient: dx = conv(round(dy), round(w)) over the flipped filter
        dx = ops.conv2d_dgrad(dy_nhwc, w_ohwi, tuple(x_nhwc.shape), st, p)
        if st == 1:
            dx_ref = torch.nn.grad.conv2d_input(x.shape, rnd(w), rnd(dy), stride=st, padding=p)
        else:   # stride-2 data gradient: fp32 kernel, unrounded operands
            dx_ref = torch.nn.grad.conv2d_input(x.shape, w, dy, stride=st, padding=p)
        close(dx.permute(0, 3, 1, 2), dx_ref, 1e-4)


def test_bf16_operand_batched_gemm():
    from mmfn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    V = torch.randn(36, 512, 128, generator=g)
    U = torch.randn(36, 256, 128, generator=g)
    out = torch.empty(36, 512, 256, device=dev)
    with ops.precision("bf16"):
        ops.gemm(V.to(dev), U.to(dev), out, 512, 256, 128, 128, 128, 256, ops.A_ROWMAJOR, ops.B_NK, batch=36,
                 strideA=512 * 128, strideB=256 * 128, strideC=512 * 256)
    ref = torch.einsum("tmk,tnk->tmn", V.bfloat16().float(), U.bfloat16().float())
    assert (out.cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_batched_gemm_split_k():
    """Packed batched GEMMs (the Winograd-domain weight gradient: 36 x [Co x Ci x tiles]) may split K."""
    from mmfn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    dM = torch.randn(36, 512, 128, generator=g)   # [t][tiles][Co]
    V = torch.randn(36, 512, 64, generator=g)     # [t][tiles][Ci]
    ref = torch.einsum("tkm,tkn->tmn", dM, V)
    for sk in (1, 4):
        out = torch.empty(36, 128, 64, device=dev)
        ops.gemm(dM.to(dev), V.to(dev), out, 128, 64, 512, 128, 64, 64, ops.A_COLMAJOR, ops.B_KN, batch=36,
                 strideA=512 * 128, strideB=512 * 64, strideC=128 * 64, tile=2, splitk=sk)
        assert (out.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item(), sk


@pytest.mark.parametrize("M,N,K", [(6144, 192, 64), (6144, 2048, 512), (6144, 64, 64), (2048, 384, 128), (256, 100, 36), (250, 64, 64)])
@pytest.mark.parametrize("tile", [0, 1, 2, 5, 7])
@pytest.mark.parametrize("splitk", [0, 1, 3])
def test_bias_gradient_from_the_weight_gradient_gemm(M, N, K, tile, splitk):
    """MMFN_EPI_COLSUM_A: linear_dw(dy, x, db=...) returns dW = dy^T x AND db = sum over rows of dy from ONE launch (+ the split-K
    combine) where the fast TN kernel runs, and through colsum() elsewhere (M not a multiple of 16): both against fp64, the
    weight gradient bit-identical to the launch without the flag."""
    from mmfn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    dy = torch.randn(M, N, generator=g)
    x = torch.randn(M, K, generator=g)
    dyd, xd = dy.to(dev), x.to(dev)
    db = torch.full((N,), float("nan"), device=dev)
    dw = ops.linear_dw(dyd, xd, db=db, tile=tile, splitk=splitk)
    _close(dw, dy.double().t() @ x.double())
    ref = dy.double().sum(0)
    err = (db.cpu().double() - ref).abs().max().item()
    assert err <= 1e-5 * (dy.abs().double().sum(0).max().item() + 1e-6), err
    assert torch.equal(dw, ops.linear_dw(dyd, xd, tile=tile, splitk=splitk))


def test_bias_gradient_from_the_weight_gradient_gemm_under_the_f32x3_table():
    """MMFN_F32X3=1 loads tuning/gfx950_f32x3.json, whose entries move the transformers' weight-gradient shapes to the three-term
    bf16 emulation kernel - which cannot return column sums (mmfn_gemm_f32: COLSUM_A + BF16X3 = MMFN_EINVAL; round 5 crashed the
    training step here).  linear_dw(db=...) must then stay on the native kernel for those shapes and still return both."""
    import os
    from mmfn_amd import ops
    dev = _dev()
    saved_flag, saved_table = ops.F32X3, dict(ops._tuned)
    try:
        ops.F32X3 = True
        ops.load_tuning(os.path.join(os.path.dirname(ops._TUNE_FILE), "gfx950_f32x3.json"))
        emulated = [k for k, v in ops._tuned.items() if len(v) == 3 and k.startswith("1,1,")]
        assert emulated, "the f32x3 table no longer holds a TN entry: pick another shape for this test"
        for key in emulated[:3]:
            N, K, M = (int(v) for v in key.split("|")[0].split(",")[2:5])
            g = torch.Generator().manual_seed(N + K)
            dy, x = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g)
            db = torch.full((N,), float("nan"), device=dev)
            dw = ops.linear_dw(dy.to(dev), x.to(dev), db=db)
            _close(dw, dy.double().t() @ x.double())
            err = (db.cpu().double() - dy.double().sum(0)).abs().max().item()
            assert err <= 1e-5 * (dy.abs().double().sum(0).max().item() + 1e-6), (key, err)
    finally:
        ops.F32X3 = saved_flag
        ops._tuned.clear()
        ops._tuned.update(saved_table)


@pytest.mark.parametrize("cfg", [(8, 64, 64, 64), (8, 32, 128, 128)])
def test_winograd_weight_gradient_sums_its_split_k_slices_in_the_output_transform(cfg, monkeypatch):
    """MMFN_EPI_KEEP_SLABS + mmfn_wino_wgrad_out_slabs_f32: the 36-batch weight-gradient GEMM (K = tiles) leaves its slices in the
    workspace and the G^T dU G transform sums them while it reads - no combine launch.  Same result as the combine + transform pair
    (to the rounding of a different summation order), bitwise repeatable, and the slab path is really the one taken."""
    from mmfn_amd import ops
    dev = _dev()
    B, HW, Cin, Cout = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, HW, HW, Cin, generator=g).to(dev)
    dy = torch.randn(B, HW, HW, Cout, generator=g).to(dev)
    taken = []
    real = ops._call

    def spy(name, *a):
        taken.append(name)
        return real(name, *a)

    monkeypatch.setattr(ops, "_call", spy)
    monkeypatch.setattr(ops, "WGRAD_SLABS", True)
    dw1 = ops.conv2d_wgrad_winograd(dy, x, torch.empty(Cout, 3, 3, Cin, device=dev))
    dw1b = ops.conv2d_wgrad_winograd(dy, x, torch.empty(Cout, 3, 3, Cin, device=dev))
    assert "mmfn_wino_wgrad_out_slabs_f32" in taken, "this shape was expected to split K"
    assert torch.equal(dw1, dw1b)
    monkeypatch.setattr(ops, "WGRAD_SLABS", False)
    dw0 = ops.conv2d_wgrad_winograd(dy, x, torch.empty(Cout, 3, 3, Cin, device=dev))
    assert (dw1 - dw0).abs().max().item() <= 2e-6 * dw0.abs().max().item()
    xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(False)
    wr = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, padding=1).backward(dy.permute(0, 3, 1, 2).double().cpu())
    _close(dw1.permute(0, 3, 1, 2).cpu(), wr.grad.float())
