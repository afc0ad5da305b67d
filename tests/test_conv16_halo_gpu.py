"""mmfn_conv3x3_halo_bf16 (csrc/conv16_halo.hip): the 3x3 stride-1 convolution of the bf16 mode over an LDS-resident halo patch
with the producer's elementwise pass in its loader.  Checked (a) against torch on the same bf16-rounded inputs, (b) BIT FOR BIT
against the two-launch path it replaces (mmfn_bn_apply_bf16 / mmfn_bn_bwd_bf16's apply -> mmfn_gemm_bf16's implicit GEMM): the k
order is the same, so the outputs must be identical, and the applied tensor it writes must be the apply kernel's."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(BF)


def _close(got, ref, tol=1.5e-2):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * scale + 1e-6, "max err %g vs scale %g" % (err, scale)


# (B, H, W, K, N): the trunk shapes of layer1-4 (small batches), the 8x8 two-images-per-tile case, a non-square map
SHAPES = [(2, 64, 64, 64, 64), (2, 32, 32, 128, 128), (4, 16, 16, 256, 256), (4, 8, 8, 512, 512), (2, 16, 32, 64, 128), (6, 8, 8, 128, 64)]


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("B,H,W,K,N", SHAPES)
def test_plain_convolution_equals_the_implicit_gemm_bit_for_bit(B, H, W, K, N, tile):
    from mmfn_amd import ops, ops16
    if tile in (2, 4) and N % 128:
        pytest.skip("128-channel tiles need N % 128 == 0")
    import ctypes
    from mmfn_amd._lib import lib
    if tile and lib().mmfn_conv3x3_halo_bf16_ok(ctypes.byref(ops16._halo_desc(B, H, W, K, N, tile=tile))) != tile:
        pytest.skip("this tile's patch does not fit the LDS for this shape (the library refuses it)")
    x, w = _rnd(B, H, W, K, seed=1), _rnd(N, 3, 3, K, scale=0.05, seed=2)
    assert ops16.halo_ok(x.shape, w.shape, 1, 1) > 0
    ref = torch.empty(B, H, W, N, dtype=BF, device=DEV)
    ops16.conv2d_fwd(x, w, 1, 1, ref)
    for stages in (0, 2, 3, 4):
        out = torch.full((B, H, W, N), float("nan"), dtype=BF, device=DEV)
        stats = torch.zeros(2 * B * H * W // 64, 2, N, dtype=torch.float64, device=DEV)
        rows = ops16.conv3x3_halo(x, w, out, stats=stats, tile=tile, stages=stages)
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), "tile %d stages %d" % (tile, stages)
        assert 0 < rows <= stats.shape[0]
        t = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(-1, N).double()
        assert torch.allclose(stats[:rows, 0].sum(0), t.sum(0), rtol=2e-3, atol=2e-3 * t.abs().sum(0).max().item())
        assert torch.allclose(stats[:rows, 1].sum(0), (t * t).sum(0), rtol=2e-3)
    _close(ref, t.view(B, H, W, N))


@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, True)])
@pytest.mark.parametrize("B,H,W,K,N", SHAPES[:4])
def test_batchnorm_apply_in_the_loader(B, H, W, K, N, relu, with_res):
    """pro 1: conv(relu(bn(co) + res)) in one launch == bn_apply then conv, and the activation it writes == bn_apply's."""
    from mmfn_amd import ops, ops16
    co, w = _rnd(B, H, W, K, seed=1, scale=2.0), _rnd(N, 3, 3, K, scale=0.05, seed=2)
    res = _rnd(B, H, W, K, seed=3) if with_res else None
    mean, rstd = torch.randn(K, device=DEV) * 0.3, torch.rand(K, device=DEV) + 0.5
    gamma, beta = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.2
    M = B * H * W
    y_ref = torch.empty(B, H, W, K, dtype=BF, device=DEV)
    ops.bn_apply(co.view(M, K), y_ref.view(M, K), mean, rstd, gamma, beta, relu, res=None if res is None else res.view(M, K))
    out_ref = torch.empty(B, H, W, N, dtype=BF, device=DEV)
    ops16.conv2d_fwd(y_ref, w, 1, 1, out_ref)
    y = torch.full((B, H, W, K), float("nan"), dtype=BF, device=DEV)
    out = torch.empty(B, H, W, N, dtype=BF, device=DEV)
    ops16.conv3x3_halo(co, w, out, bn_apply=(mean, rstd, gamma, beta, res, relu, y))
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
    assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16))
    # and against torch
    t = co.float() * (gamma * rstd) + (beta - mean * gamma * rstd)
    if res is not None:
        t = t + res.float()
    if relu:
        t = torch.relu(t)
    _close(y, t, tol=1e-2)


@pytest.mark.parametrize("B,H,W,K,N", SHAPES[:4])
def test_data_gradient_with_skip_gradient_and_emitted_reductions(B, H, W, K, N):
    """flip: dx = conv_transpose(dco) over the [Cin,3,3,Cout] shadow (+ the skip branch's gradient) with the BatchNorm-backward
    reductions of the layer below from the epilogue == mmfn_gemm_bf16's MMFN_G16_CONV_DGRAD with stats_mode 2."""
    from mmfn_amd import ops, ops16
    # here K = Cout of the convolution (channels of dco), N = Cin (channels of dx)
    dco, w_t = _rnd(B, H, W, K, seed=1), _rnd(N, 3, 3, K, scale=0.05, seed=2)
    res = _rnd(B, H, W, N, seed=3)
    y_c, x_c = _rnd(B, H, W, N, seed=4), _rnd(B, H, W, N, seed=5, scale=2.0)
    mean, rstd = torch.randn(N, device=DEV) * 0.1, torch.rand(N, device=DEV) + 0.5
    Mx = B * H * W
    part_ref = torch.zeros(ops16.max_stats_rows(Mx), 2, N, dtype=torch.float64, device=DEV)
    dx_ref = torch.empty(B, H, W, N, dtype=BF, device=DEV)
    ops16.conv2d_dgrad(dco, w_t, (B, H, W, N), (K, 3, 3, N), 1, 1, dx_ref, res=res.view(-1, N), ldr=N, stats=part_ref, stats_mode=2,
                       bn=(y_c, x_c, mean, rstd))
    part = torch.zeros_like(part_ref)
    dx = torch.empty_like(dx_ref)
    rows = ops16.conv3x3_halo(dco, w_t, dx, flip=True, out_res=res, stats=part, stats_mode=2, bn2=(y_c, x_c, mean, rstd))
    assert torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16))
    assert torch.allclose(part[:rows].sum(0), part_ref.sum(0), rtol=1e-5, atol=1e-5 * float(part_ref.sum(0).abs().max()))
    xt = torch.zeros(B, N, H, W, device=DEV, requires_grad=True)
    F.conv2d(xt, w_t.float().permute(3, 0, 1, 2), padding=1).backward(dco.float().permute(0, 3, 1, 2))
    _close(dx, xt.grad.permute(0, 2, 3, 1) + res.float())


@pytest.mark.parametrize("masked", [True, False])
@pytest.mark.parametrize("B,H,W,K,N", SHAPES[:4])
def test_batchnorm_backward_in_the_loader(B, H, W, K, N, masked):
    """pro 2: the data gradient of a convolution whose output gradient is the BatchNorm backward of g, formed while the patch is
    staged; dco / ge written for the owned pixels == mmfn_bn_bwd_bf16 (within one bf16 rounding: the compiler may contract the
    two kernels' expressions differently), dx == the implicit GEMM over the dco this launch wrote, bit for bit."""
    from mmfn_amd import ops, ops16
    g, co = _rnd(B, H, W, K, seed=1), _rnd(B, H, W, K, seed=2, scale=2.0)
    y = _rnd(B, H, W, K, seed=3) if masked else None
    w_t = _rnd(N, 3, 3, K, scale=0.05, seed=4)
    mean, rstd = torch.randn(K, device=DEV) * 0.1, torch.rand(K, device=DEV) + 0.5
    gamma = torch.rand(K, device=DEV) + 0.5
    M = B * H * W
    dco_ref, ge_ref = torch.empty(M, K, dtype=BF, device=DEV), torch.empty(M, K, dtype=BF, device=DEV)
    dwt, dbs = torch.empty(K, device=DEV), torch.empty(K, device=DEV)
    ops.bn_bwd(g.view(M, K), None if y is None else y.view(M, K), co.view(M, K), mean, rstd, gamma, dco_ref, dwt, dbs, ge_out=ge_ref)
    # the means the apply pass used: mean(ge), mean(ge * xhat) = dbias / M, dweight / M
    means = torch.stack([dbs / M, dwt / M]).contiguous()
    dco, ge = torch.full((M, K), float("nan"), dtype=BF, device=DEV), torch.full((M, K), float("nan"), dtype=BF, device=DEV)
    dx = torch.empty(B, H, W, N, dtype=BF, device=DEV)
    ops16.conv3x3_halo(g, w_t, dx, flip=True, bn_bwd=(mean, rstd, gamma, means, y, co, dco, ge))
    assert torch.equal(ge.view(torch.int16), ge_ref.view(torch.int16))
    d = (dco.float() - dco_ref.float()).abs()
    assert float((d / dco_ref.float().abs().clamp_min(1e-3)).max()) <= 2.0 ** -7     # at most one bf16 ulp apart
    assert float((d > 0).float().mean()) < 0.05
    dx_ref = torch.empty_like(dx)
    ops16.conv2d_dgrad(dco.view(B, H, W, K), w_t, (B, H, W, N), (K, 3, 3, N), 1, 1, dx_ref)
    assert torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16))


def test_shapes_the_kernel_does_not_serve_are_refused():
    from mmfn_amd import ops16
    assert ops16.halo_ok((1, 8, 8, 512), (512, 3, 3, 512), 1, 1) > 0          # one image: 64-pixel tiles
    assert ops16.halo_ok((2, 64, 64, 64), (64, 3, 3, 64), 2, 1) == 0          # strided
    assert ops16.halo_ok((2, 64, 64, 64), (64, 1, 1, 64), 1, 0) == 0          # 1x1
    assert ops16.halo_ok((2, 60, 64, 64), (64, 3, 3, 64), 1, 1) == 0          # not a power of two
    assert ops16.halo_ok((2, 64, 64, 32), (64, 3, 3, 32), 1, 1) == 0          # channels
