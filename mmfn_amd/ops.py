"""Thin host wrappers: torch tensors (device memory only) -> C-ABI launches.

Every function enqueues HIP kernels from libmmfn_hip.so on the current torch stream and
returns without synchronising.  Feature maps are NHWC ([B,H,W,C] contiguous); conv weights are
[Cout,KH,KW,Cin] contiguous; Linear weights are [out,in] as in the reference checkpoint.
"""
import ctypes

import torch

from . import _lib
from ._lib import (A_COLMAJOR, A_DGRAD, A_IM2COL, A_ROWMAJOR, B_DGRADW, B_IM2COL, B_KN, B_NK,
                   EPI_ACCUM, EPI_BIAS, EPI_DROPOUT, EPI_GELU, EPI_MASK_AUX, EPI_RELU, EPI_RESIDUAL,
                   GemmDesc, check, lib, ptr, stream)

_workspace = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer (split-K slabs); never freed so captured graphs stay valid."""
    key = str(device)
    buf = _workspace.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("split-K workspace must be sized before graph capture")
        buf = torch.empty(max(nbytes // 4 + 1, 1 << 22), dtype=torch.float32, device=device)
        _workspace[key] = buf
    return buf


def _f32c(t, name):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda, name
    return t


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_mode=A_ROWMAJOR, b_mode=B_NK, bias=None, res=None, ldr=0,
         aux=None, ldaux=0, relu=False, gelu=False, accum=False, drop_p=0.0, rng_state=None, rng_stream=0,
         conv=None, splitk=0, tile=0):
    d = GemmDesc()
    d.A, d.B, d.C = ptr(A), ptr(B), ptr(C)
    d.bias, d.res, d.aux = ptr(bias), ptr(res), ptr(aux)
    d.rng_state = ptr(rng_state)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldr, d.ldaux = lda, ldb, ldc, ldr, ldaux
    d.a_mode, d.b_mode = a_mode, b_mode
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
        d.drop_p = drop_p
        d.rng_stream = rng_stream
    if res is not None:
        flags |= EPI_RESIDUAL
    if accum:
        flags |= EPI_ACCUM
    d.flags = flags
    d.splitk = splitk
    d.tile = tile
    L = lib()
    need = L.mmfn_gemm_workspace_bytes(ctypes.byref(d))
    if need > 0:
        d.workspace = ptr(workspace(need, C.device))
    check(L.mmfn_gemm_f32(ctypes.byref(d), stream()), "mmfn_gemm_f32")
    return C


# ---------------------------------------------------------------- Linear
def linear_fwd(x, w, bias=None, out=None, **epi):
    """out[M,N] = x[M,K] @ w[N,K]^T (+bias, epilogue)."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    return gemm(x, w, out, M, N, K, x.stride(0), w.stride(0), out.stride(0), A_ROWMAJOR, B_NK, bias=bias, **epi)


def linear_dx(dy, w, out=None, **epi):
    """dx[M,K] = dy[M,N] @ w[N,K]."""
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty(M, K, dtype=torch.float32, device=dy.device)
    return gemm(dy, w, out, M, K, N, dy.stride(0), w.stride(0), out.stride(0), A_ROWMAJOR, B_KN, **epi)


def linear_dw(dy, x, out=None, **epi):
    """dw[N,K] = dy[M,N]^T @ x[M,K]."""
    M, N = dy.shape
    K = x.shape[1]
    if out is None:
        out = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    return gemm(dy, x, out, N, K, M, dy.stride(0), x.stride(0), out.stride(0), A_COLMAJOR, B_KN, **epi)


# ---------------------------------------------------------------- Convolution (NHWC)
def conv_geom(x_shape, w_shape, stride, pad):
    B, H, W, Cin = x_shape
    Cout, KH, KW, Cin2 = w_shape
    assert Cin == Cin2
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    return (H, W, Cin, OH, OW, Cout, KH, KW, stride, pad), (B, OH, OW, Cout)


def conv2d_fwd(x, w, stride, pad, out=None, **epi):
    """y[B,OH,OW,Cout] = conv(x[B,H,W,Cin], w[Cout,KH,KW,Cin])."""
    g, oshape = conv_geom(x.shape, w.shape, stride, pad)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    B, OH, OW, Cout = oshape
    K = g[6] * g[7] * g[2]
    return gemm(x, w, out, B * OH * OW, Cout, K, 0, K, Cout, A_IM2COL, B_NK, conv=g, **epi)


def conv2d_dgrad(dy, w, x_shape, stride, pad, out=None, **epi):
    """dx[B,H,W,Cin] from dy[B,OH,OW,Cout]."""
    g, oshape = conv_geom(x_shape, w.shape, stride, pad)
    assert tuple(dy.shape) == oshape
    if out is None:
        out = torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    B, H, W, Cin = x_shape
    K = g[6] * g[7] * g[5]
    return gemm(dy, w, out, B * H * W, Cin, K, 0, 0, Cin, A_DGRAD, B_DGRADW, conv=g, **epi)


def conv2d_wgrad(dy, x, w_shape, stride, pad, out=None, **epi):
    """dw[Cout,KH,KW,Cin] = sum over pixels of dy (x) im2col(x)."""
    g, oshape = conv_geom(x.shape, w_shape, stride, pad)
    assert tuple(dy.shape) == oshape
    if out is None:
        out = torch.empty(w_shape, dtype=torch.float32, device=dy.device)
    B, OH, OW, Cout = oshape
    N = g[6] * g[7] * g[2]
    return gemm(dy, x, out, Cout, N, B * OH * OW, Cout, 0, N, A_COLMAJOR, B_IM2COL, conv=g, **epi)
