"""Host wrappers of the bf16 training mode (BASELINE configs[2]): torch tensors (device memory only) -> C-ABI launches.

Activations, saved tensors and the per-step weight shadows are bf16 in HBM; master weights, gradients, normalisation
statistics, the loss head and the optimizer stay fp32 (mmfn_amd.ops).  Same conventions as mmfn_amd.ops: NHWC feature maps,
[rows, C] token matrices, everything enqueued on the current torch stream, nothing synchronises.
"""
import ctypes

import torch

from . import ops
from ._lib import (EPI16_OUT_F32, EPI_ACCUM, EPI_BIAS, EPI_DROPOUT, EPI_GELU, EPI_MASK_AUX, EPI_RELU, EPI_RELU_LAST, EPI_RESIDUAL,
                   G16_CONV_DGRAD, G16_CONV_FWD, G16_CONV_WGRAD, G16_NT, G16_TN, Gemm16Desc, check, lib, ptr, stream)

BF16 = torch.bfloat16
_profiler_tag = "bf16 "


def gemm16(form, A, B, C, M, N, K, lda, ldb, ldc, bias=None, res=None, ldr=0, aux=None, ldaux=0, relu=False, gelu=False, accum=False,
           relu_last=False, drop_p=0.0, rng_state=None, rng_stream=0, conv=None, stats=None, tile=0, splitk=0):
    """One launch of mmfn_gemm_bf16 (include/mmfn_hip.h).  C dtype decides MMFN_EPI16_OUT_F32."""
    d = Gemm16Desc()
    d.A, d.B, d.C = ptr(A), ptr(B), ptr(C)
    d.bias, d.res, d.aux, d.rng_state, d.stats = ptr(bias), ptr(res), ptr(aux), ptr(rng_state), ptr(stats)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldr, d.ldaux = lda, ldb, ldc, ldr, ldaux
    d.form = form
    if conv is not None:
        (d.H, d.W, d.Cin, d.OH, d.OW, d.Cout, d.KH, d.KW, d.stride, d.pad) = conv
    flags = 0
    if bias is not None:
        flags |= EPI_BIAS
    if relu:
        flags |= EPI_RELU
    if gelu:
        flags |= EPI_GELU
    if aux is not None:
        flags |= EPI_MASK_AUX
    if drop_p > 0.0:
        flags |= EPI_DROPOUT
        d.drop_p, d.rng_stream = drop_p, rng_stream
    if res is not None:
        flags |= EPI_RESIDUAL
    if accum:
        flags |= EPI_ACCUM
    if relu_last:
        flags |= EPI_RELU_LAST
    if C.dtype == torch.float32:
        flags |= EPI16_OUT_F32
    else:
        assert C.dtype == BF16
    assert A.dtype == BF16 and B.dtype == BF16
    d.flags, d.tile, d.splitk = flags, tile, splitk
    L = lib()
    need = L.mmfn_gemm_bf16_workspace_bytes(ctypes.byref(d))
    if need > 0:
        d.workspace = ptr(ops.workspace(need, C.device))
    prof = ops._profiler
    if prof is None or prof.suspended:
        check(L.mmfn_gemm_bf16(ctypes.byref(d), stream()), "mmfn_gemm_bf16")
        return C
    if conv is not None:
        H, W, Cin, OH, OW, Cout, KH, KW, st, pd = conv
        nb = (M // (H * W)) if form == G16_CONV_DGRAD else ((M // (OH * OW)) if form == G16_CONV_FWD else (K // (OH * OW)))
        flops = 2.0 * nb * OH * OW * Cin * KH * KW * Cout
        abytes = 2.0 * (nb * H * W * Cin + nb * OH * OW * Cout) + (4.0 if form == G16_CONV_WGRAD else 2.0) * Cout * KH * KW * Cin
        tag = "%s %dx%d c%d->%d k%d s%d" % ({G16_CONV_FWD: "conv", G16_CONV_DGRAD: "dgrad", G16_CONV_WGRAD: "wgrad"}[form], H, W, Cin, Cout, KH, st)
    else:
        flops = 2.0 * M * N * K
        abytes = 2.0 * (M * K + N * K) + C.element_size() * M * N
        tag = "gemm16 f%d %dx%dx%d" % (form, M, N, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(L.mmfn_gemm_bf16(ctypes.byref(d), stream()), "mmfn_gemm_bf16")
    e1.record()
    prof.records.append((e0, e1, flops, _profiler_tag + tag, abytes, flops))
    return C


def stats_rows(M, N, K, form=G16_CONV_FWD, tile=0):
    d = Gemm16Desc()
    d.M, d.N, d.K, d.form, d.tile = M, N, K, form, tile
    return lib().mmfn_gemm_bf16_stats_rows(ctypes.byref(d))


# ---------------------------------------------------------------- Linear
def linear_fwd(x, w16, bias=None, out=None, **epi):
    """out[M,N] = x[M,K] @ w16[N,K]^T (+ epilogue); x, w16 bf16."""
    M, K = x.shape
    N = w16.shape[0]
    return gemm16(G16_NT, x, w16, out, M, N, K, x.stride(0), w16.stride(0), out.stride(0), bias=bias, **epi)


def linear_dx(dy, w16t, out=None, **epi):
    """dx[M,K] = dy[M,N] @ w[N,K], with w16t = w^T stored [K,N] (the transposed bf16 shadow)."""
    M, N = dy.shape
    K = w16t.shape[0]
    return gemm16(G16_NT, dy, w16t, out, M, K, N, dy.stride(0), w16t.stride(0), out.stride(0), **epi)


def linear_dw(dy, x, out, **epi):
    """dw[N,K] (fp32) = dy[M,N]^T @ x[M,K]; dy, x bf16."""
    M, N = dy.shape
    K = x.shape[1]
    return gemm16(G16_TN, dy, x, out, N, K, M, dy.stride(0), x.stride(0), out.stride(0), **epi)


# ---------------------------------------------------------------- Convolution (NHWC bf16)
def conv2d_fwd(x, w16, stride, pad, out, stats=None, **epi):
    """y[B,OH,OW,Cout] = conv(x[B,H,W,Cin], w16[Cout,KH,KW,Cin]); stats: optional fp64 [rows,2,Cout] BatchNorm partial sums."""
    g, oshape = ops.conv_geom(x.shape, w16.shape, stride, pad)
    B, OH, OW, Cout = oshape
    K = g[6] * g[7] * g[2]
    return gemm16(G16_CONV_FWD, x, w16, out, B * OH * OW, Cout, K, 0, K, Cout, conv=g, stats=stats, **epi)


def conv2d_dgrad(dy, w16t, x_shape, w_shape, stride, pad, out, **epi):
    """dx[B,H,W,Cin] from dy[B,OH,OW,Cout]; w16t = the [Cin,KH,KW,Cout] shadow of w[Cout,KH,KW,Cin]."""
    g, oshape = ops.conv_geom(x_shape, w_shape, stride, pad)
    assert tuple(dy.shape) == oshape
    B, H, W, Cin = x_shape
    K = g[6] * g[7] * g[5]
    return gemm16(G16_CONV_DGRAD, dy, w16t, out, B * H * W, Cin, K, 0, K, Cin, conv=g, **epi)


def conv2d_wgrad(dy, x, w_shape, stride, pad, out, **epi):
    """dw[Cout,KH,KW,Cin] (fp32) = sum over pixels of dy (x) im2col(x); dy, x bf16."""
    g, oshape = ops.conv_geom(x.shape, w_shape, stride, pad)
    assert tuple(dy.shape) == oshape
    B, OH, OW, Cout = oshape
    N = g[6] * g[7] * g[2]
    return gemm16(G16_CONV_WGRAD, dy, x, out, Cout, N, B * OH * OW, Cout, 0, N, conv=g, **epi)
