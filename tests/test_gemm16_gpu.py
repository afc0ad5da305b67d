"""mmfn_gemm_bf16 (bf16 operands in HBM, v_mfma_f32_32x32x16_bf16, fp32 accumulate) against torch on the same bf16-rounded
inputs: every form (Linear forward / data gradient / weight gradient, convolution forward / data gradient of stride 1 and 2 /
weight gradient), every tile shape, the epilogue flags, the BatchNorm statistics by-product."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(torch.bfloat16)


def _close(got, ref, tol=1.5e-2):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * scale + 1e-6, "max err %g vs scale %g" % (err, scale)


@pytest.mark.parametrize("stages", [2, 3, 4])
def test_lds_pipeline_depths(stages):
    """2 = double buffer, 3 / 4 = one / two k-tiles in flight beyond it (counted vmcnt waits): same results, K = 1 .. 9 tiles."""
    from mmfn_amd import ops16
    for K in (64, 128, 192, 576):
        x, w = _rnd(300, K, seed=1), _rnd(136, K, scale=0.1, seed=2)
        out = torch.empty(300, 136, dtype=torch.float32, device=DEV)
        ops16.linear_fwd(x, w, None, out=out, stages=stages, tile=2)
        _close(out, x.float() @ w.float().t(), tol=2e-3)
        dw = torch.empty(136, K, dtype=torch.float32, device=DEV)
        dy = _rnd(300, 136, seed=3)
        ops16.linear_dw(dy, x, dw, stages=stages, tile=2, splitk=1)
        _close(dw, dy.float().t() @ x.float(), tol=2e-3)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(6144, 512, 512), (200, 64, 64), (6144, 192, 256), (2048, 2048, 512)])
def test_linear_forward_and_dx(M, N, K, tile):
    from mmfn_amd import ops16
    x, w = _rnd(M, K, seed=1), _rnd(N, K, scale=0.05, seed=2)
    bias = torch.randn(N, device=DEV)
    res = _rnd(M, N, seed=3)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops16.linear_fwd(x, w, bias, out=out, res=res, ldr=N, relu=True, tile=tile)
    ref = torch.relu(x.float() @ w.float().t() + bias) + res.float()
    _close(out, ref)
    out32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops16.linear_fwd(x, w, None, out=out32, tile=tile)
    _close(out32, x.float() @ w.float().t(), tol=2e-3)
    # data gradient over the transposed shadow, ReLU mask of the layer below in the epilogue
    dy = _rnd(M, N, seed=4)
    wt = w.t().contiguous()                      # [K, N]
    aux = _rnd(M, K, seed=5)
    dx = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    ops16.linear_dx(dy, wt, out=dx, aux=aux, ldaux=K, tile=tile)
    ref = (dy.float() @ w.float()) * (aux.float() > 0)
    _close(dx, ref)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(6144, 512, 2048), (6144, 1536, 512), (520, 64, 256), (64, 128, 64)])
def test_linear_weight_gradient_transpose_reads(M, N, K, tile):
    """dW = dY^T X through ds_read_b64_tr_b16 (both operands contraction-major): asymmetric operands, ragged contraction."""
    from mmfn_amd import ops16
    dy, x = _rnd(M, N, seed=6), _rnd(M, K, seed=7)
    dw = torch.empty(N, K, dtype=torch.float32, device=DEV)
    ops16.linear_dw(dy, x, dw, tile=tile)
    _close(dw, dy.float().t() @ x.float(), tol=2e-3)
    for sk in (1, 3):
        ops16.linear_dw(dy, x, dw, tile=tile, splitk=sk)
        _close(dw, dy.float().t() @ x.float(), tol=2e-3)


def test_dropout_epilogue_matches_the_f32_kernels_mask():
    from mmfn_amd import ops, ops16
    M, N, K = 512, 256, 128
    x, w = _rnd(M, K, seed=1), _rnd(N, K, scale=0.1, seed=2)
    rng = torch.tensor([1234, 7], dtype=torch.int64, device=DEV)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops16.linear_fwd(x, w, None, out=out, drop_p=0.25, rng_state=rng, rng_stream=11)
    ones = torch.ones(M, N, device=DEV)
    mask = ops.dropout_apply(ones, torch.empty_like(ones), 0.25, rng, 11)   # the fp32 path's mask for the same (state, stream)
    _close(out, (x.float() @ w.float().t()) * mask)


CONVS = [(2, 32, 32, 64, 64, 3, 1, 1), (2, 16, 16, 128, 256, 3, 2, 1), (3, 16, 16, 128, 256, 1, 2, 0), (2, 8, 8, 512, 512, 3, 1, 1),
         (2, 64, 64, 64, 128, 3, 2, 1), (4, 16, 16, 256, 256, 3, 1, 1)]


@pytest.mark.parametrize("B,H,W,Ci,Co,k,s,p", CONVS)
def test_convolution_forward_dgrad_wgrad(B, H, W, Ci, Co, k, s, p):
    from mmfn_amd import ops, ops16
    x = _rnd(B, H, W, Ci, seed=1)
    w = _rnd(Co, k, k, Ci, scale=0.05, seed=2)
    g, oshape = ops.conv_geom(x.shape, w.shape, s, p)
    xt, wt = x.float().permute(0, 3, 1, 2).requires_grad_(True), w.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv2d(xt, wt, stride=s, padding=p)
    y = torch.empty(oshape, dtype=torch.bfloat16, device=DEV)
    rows = ops16.conv_stats_rows(x.shape, w.shape, s, p)   # partial rows of the tile the tuning table picks for this shape
    stats = torch.zeros(rows, 2, Co, dtype=torch.float64, device=DEV)
    ops16.conv2d_fwd(x, w, s, p, y, stats=stats)
    _close(y, ref.permute(0, 2, 3, 1))
    # BatchNorm statistics by-product: sums of the fp32 accumulators over all output pixels
    ref2 = ref.detach().permute(0, 2, 3, 1).reshape(-1, Co).double()
    assert torch.allclose(stats[:, 0].sum(0), ref2.sum(0), rtol=2e-3, atol=2e-3 * ref2.abs().sum(0).max().item())
    assert torch.allclose(stats[:, 1].sum(0), (ref2 * ref2).sum(0), rtol=2e-3)
    dy = _rnd(*oshape, seed=3)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    w_t = w.permute(3, 1, 2, 0).contiguous()      # [Ci, kh, kw, Co] shadow
    dx = torch.empty_like(x)
    ops16.conv2d_dgrad(dy, w_t, tuple(x.shape), tuple(w.shape), s, p, dx)
    _close(dx, xt.grad.permute(0, 2, 3, 1))
    dw = torch.empty(Co, k, k, Ci, dtype=torch.float32, device=DEV)
    ops16.conv2d_wgrad(dy, x, tuple(w.shape), s, p, dw)
    _close(dw, wt.grad.permute(0, 2, 3, 1), tol=3e-3)


@pytest.mark.parametrize("stride", [1, 2])   # 2: the data gradient runs in parity-pure row tiles (gemm_bf16.hip parity_pixel_row)
def test_epilogue_partial_sums_bias_gradient_and_batchnorm_backward_reductions(stride):
    """stats_mode 1: column sums of the FINAL epilogue value (the bias gradient of the Linear whose dX the launch computes);
    stats_mode 2: the two reductions of the BatchNorm backward that the launch's output gradient enters, with the consumer's
    ReLU mask - against mmfn_bn_bwd_bf16's own reduction pass on the stored tensors."""
    from mmfn_amd import ops, ops16
    M, N, K = 6144, 512, 128
    dy, wt, aux = _rnd(M, K, seed=1), _rnd(N, K, scale=0.1, seed=2), _rnd(M, N, seed=3)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    part = torch.zeros(ops16.max_stats_rows(M), 2, N, dtype=torch.float64, device=DEV)
    ops16.linear_dx(dy, wt, out=out, aux=aux, ldaux=N, stats=part, stats_mode=1)
    rows = ops16.gemm_stats_rows(ops16.G16_NT, M, N, K)
    got = ops16.colsum_partials(part, rows, N, torch.empty(N, device=DEV))
    ref = ((dy.float() @ wt.float().t()) * (aux.float() > 0)).sum(0)
    assert torch.allclose(got, ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
    # BatchNorm-backward reductions from a data-gradient epilogue
    B, H, W, Ci, Co = 2, 16, 16, 128, 128
    dyc = _rnd(B, H // stride, W // stride, Co, seed=4)
    w_t = _rnd(Ci, 3, 3, Co, scale=0.05, seed=5)
    res = _rnd(B, H, W, Ci, seed=6)
    y_c, x_c = _rnd(B, H, W, Ci, seed=7), _rnd(B, H, W, Ci, seed=8, scale=2.0)
    mean, rstd = torch.randn(Ci, device=DEV) * 0.1, torch.rand(Ci, device=DEV) + 0.5
    Mx = B * H * W
    part = torch.zeros(ops16.max_stats_rows(Mx), 2, Ci, dtype=torch.float64, device=DEV)
    dx = torch.empty(B, H, W, Ci, dtype=torch.bfloat16, device=DEV)
    ops16.conv2d_dgrad(dyc, w_t, (B, H, W, Ci), (Co, 3, 3, Ci), stride, 1, dx, res=res.view(-1, Ci), ldr=Ci, stats=part, stats_mode=2,
                       bn=(y_c, x_c, mean, rstd))
    # the data gradient itself (+ residual) against torch
    xt = torch.zeros(B, Ci, H, W, device=DEV, requires_grad=True)
    F.conv2d(xt, w_t.float().permute(3, 0, 1, 2), stride=stride, padding=1).backward(dyc.float().permute(0, 3, 1, 2))
    _close(dx, xt.grad.permute(0, 2, 3, 1) + res.float())
    g_, _ = ops.conv_geom((B, H, W, Ci), (Co, 3, 3, Ci), stride, 1)
    rows = ops16.gemm_stats_rows(ops16.G16_CONV_DGRAD, Mx, Ci, 9 * Co, g_)
    wgt = torch.rand(Ci, device=DEV) + 0.5
    outs = []
    for fused in (False, True):
        dco, ge = torch.empty(Mx, Ci, dtype=torch.bfloat16, device=DEV), torch.empty(Mx, Ci, dtype=torch.bfloat16, device=DEV)
        dwt, dbs = torch.empty(Ci, device=DEV), torch.empty(Ci, device=DEV)
        if fused:
            ops16.bn_bwd_partials(part, rows, dx.view(Mx, Ci), y_c.view(Mx, Ci), x_c.view(Mx, Ci), mean, rstd, wgt, dco, dwt, dbs, ge_out=ge)
        else:
            ops.bn_bwd(dx.view(Mx, Ci), y_c.view(Mx, Ci), x_c.view(Mx, Ci), mean, rstd, wgt, dco, dwt, dbs, ge_out=ge)
        outs.append((dco, ge, dwt, dbs))
    a, b = outs
    assert torch.allclose(a[2], b[2], rtol=1e-4, atol=1e-4 * float(a[2].abs().max()))
    assert torch.allclose(a[3], b[3], rtol=1e-4, atol=1e-4 * float(a[3].abs().max()))
    assert torch.equal(a[1], b[1])
    assert float((a[0].float() - b[0].float()).abs().max()) <= 2e-2 * float(a[0].float().abs().max())
