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
_lane = 0  # logical execution lane (0 = main stream, 1.. = engine side streams); scratch buffers are per lane


class lane(object):
    """Context manager: kernels launched inside use the scratch buffers of logical lane `i`.  (Keying by the
    raw stream pointer would break under hipGraph capture, where the capturing stream is a fresh one.)"""

    def __init__(self, i):
        self.i = i

    def __enter__(self):
        global _lane
        self.prev, _lane = _lane, self.i

    def __exit__(self, *exc):
        global _lane
        _lane = self.prev


class GemmProfiler(object):
    """Brackets every GEMM/conv launch with HIP events on the launch stream and tallies its
    ALGORITHMIC FLOPs (true conv/matmul FLOPs, not the zero-inflated implicit-GEMM count)."""

    def __init__(self):
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        return len(self.records), float(sum(r[2] for r in self.records)), ms

    def algo_bytes(self):
        return float(sum(r[4] for r in self.records))

    def by_tag(self):
        torch.cuda.synchronize()
        out = {}
        for s, e, f, tag, _ in self.records:
            n, fl, ms = out.get(tag, (0, 0.0, 0.0))
            out[tag] = (n + 1, fl + f, ms + s.elapsed_time(e))
        return out


_profiler = None


def set_gemm_profiler(p):
    global _profiler
    _profiler = p


def workspace(nbytes, device):
    """Grow-only scratch buffer (split-K slabs); never freed so captured graphs stay valid.
    One buffer per (device, lane): branches running concurrently on side streams must not share slabs."""
    key = (str(device), _lane)
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
         conv=None, splitk=0, tile=0, batch=1, strideA=0, strideB=0, strideC=0):
    d = GemmDesc()
    d.batch, d.strideA, d.strideB, d.strideC = batch, strideA, strideB, strideC
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
    if _profiler is None:
        check(L.mmfn_gemm_f32(ctypes.byref(d), stream()), "mmfn_gemm_f32")
        return C
    if a_mode == A_DGRAD:  # true transposed-conv FLOPs: output pixels x Cin x taps x Cout
        H, W, Cin, OH, OW, Cout, KH, KW, st, pd = conv
        flops = 2.0 * (M // (H * W)) * OH * OW * Cin * KH * KW * Cout
        tag = "dgrad %dx%d c%d->%d k%d s%d" % (H, W, Cin, Cout, KH, st)
    else:
        flops = 2.0 * M * N * K * max(1, batch)
        if conv is not None:
            H, W, Cin, OH, OW, Cout, KH, KW, st, pd = conv
            tag = "%s %dx%d c%d->%d k%d s%d" % ("conv" if a_mode == A_IM2COL else "wgrad", H, W, Cin, Cout, KH, st)
        else:
            tag = "gemm a%d b%d %dx%dx%d" % (a_mode, b_mode, M, N, K)
    # algorithmic (compulsory) bytes: every operand element read once, every output written once
    if conv is not None:
        H, W, Cin, OH, OW, Cout, KH, KW, st, pd = conv
        nb = (M // (H * W)) if a_mode == A_DGRAD else ((M // (OH * OW)) if a_mode == A_IM2COL else (K // (OH * OW)))
        abytes = 4.0 * (nb * H * W * Cin + nb * OH * OW * Cout + Cout * KH * KW * Cin)
    else:
        abytes = 4.0 * (M * K + N * K + M * N) * max(1, batch)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(L.mmfn_gemm_f32(ctypes.byref(d), stream()), "mmfn_gemm_f32")
    e1.record()
    _profiler.records.append((e0, e1, flops, tag, abytes))
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


_wflip = {}
FLIPPED_DGRAD = True  # A/B switch (tools/gemm_bench.py)


def _flip_scratch(n, device):
    """Per-(device, lane) buffer for the flipped filter of the running dgrad (grow-only; sized before graph capture)."""
    key = (str(device), _lane)
    buf = _wflip.get(key)
    if buf is None or buf.numel() < n:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("flipped-filter scratch must be sized before graph capture")
        buf = torch.empty(max(n, 512 * 9 * 512), dtype=torch.float32, device=device)
        _wflip[key] = buf
    return buf


def conv2d_dgrad(dy, w, x_shape, stride, pad, out=None, **epi):
    """dx[B,H,W,Cin] from dy[B,OH,OW,Cout]."""
    g, oshape = conv_geom(x_shape, w.shape, stride, pad)
    assert tuple(dy.shape) == oshape
    if out is None:
        out = torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    Co, KH, KW, Ci = w.shape
    if FLIPPED_DGRAD and stride == 1 and KH == KW and 2 * pad == KH - 1 and Co % 16 == 0 and Ci % 4 == 0:
        # stride 1, "same" padding: dx = conv(dy, flip(w)^T) - the forward implicit GEMM with k-contiguous weights
        # runs ~15 % faster than the gather form below; the 0.1-9 MB filter transpose costs 2-4 us
        wt = _flip_scratch(w.numel(), dy.device)[:w.numel()].view(Ci, KH, KW, Co)
        _call("mmfn_conv_weight_flip_f32", ptr(w), ptr(wt), Co, KH * KW, Ci, stream())
        return conv2d_fwd(dy, wt, 1, pad, out=out, **epi)
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


# ---------------------------------------------------------------- small helpers
def _call(name, *args):
    check(getattr(lib(), name)(*args), name)


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


_norm_ws = {}


def norm_workspace(device, nbytes=None):
    """Scratch for norm / colsum / token-bwd reductions (grow-only, fp64-aligned), per (device, lane)."""
    key = (str(device), _lane)
    need = max(nbytes or 0, lib().mmfn_norm_workspace_bytes(1024) + 8192)
    buf = _norm_ws.get(key)
    if buf is None or buf.numel() * 8 < need:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("norm workspace must be sized before graph capture")
        buf = torch.empty(need // 8 + 1, dtype=torch.float64, device=device)
        _norm_ws[key] = buf
    return buf


def fill(t, value):
    _call("mmfn_fill_f32", ptr(t), float(value), t.numel(), stream())
    return t


def axpby(y, x, a=1.0, b=1.0):
    _call("mmfn_axpby_f32", ptr(y), ptr(x), float(a), float(b), y.numel(), stream())
    return y


# ---------------------------------------------------------------- BatchNorm / LayerNorm
def bn_train_stats(x2d, mean, rstd, running_mean, running_var, nbt, eps=1e-5, momentum=0.1):
    M, C = x2d.shape
    _call("mmfn_bn_train_stats_f32", ptr(x2d), M, C, eps, momentum, ptr(mean), ptr(rstd), ptr(running_mean),
          ptr(running_var), ptr(nbt), ptr(norm_workspace(x2d.device)), stream())


def bn_eval_prepare(running_mean, running_var, mean, rstd, eps=1e-5):
    _call("mmfn_bn_eval_prepare_f32", ptr(running_mean), ptr(running_var), eps, mean.numel(), ptr(mean), ptr(rstd), stream())


def bn_apply(x2d, y2d, mean, rstd, weight, bias, relu, res=None):
    M, C = x2d.shape
    _call("mmfn_bn_apply_f32", ptr(x2d), ptr(res), ptr(y2d), M, C, ptr(mean), ptr(rstd), ptr(weight), ptr(bias),
          1 if relu else 0, stream())
    return y2d


def bn_bwd(g2d, y2d, x2d, mean, rstd, weight, dx, dweight, dbias, ge_out=None):
    M, C = x2d.shape
    _call("mmfn_bn_bwd_f32", ptr(g2d), ptr(y2d), ptr(x2d), M, C, ptr(mean), ptr(rstd), ptr(weight), ptr(dx), ptr(ge_out),
          ptr(dweight), ptr(dbias), ptr(norm_workspace(x2d.device)), stream())
    return dx


ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


def layernorm_fwd(x, w, b, y, mean, rstd, act=ACT_NONE, eps=1e-5):
    M, C = x.shape
    _call("mmfn_layernorm_fwd_f32", ptr(x), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), M, C, eps, act, stream())
    return y


def layernorm_bwd(g, x, w, b, mean, rstd, dx, dw, db, act=ACT_NONE, dres=None):
    M, C = x.shape
    _call("mmfn_layernorm_bwd_f32", ptr(g), ptr(x), ptr(w), ptr(b), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dw),
          ptr(db), M, C, act, ptr(norm_workspace(x.device)), stream())
    return dx


def colsum(x2d, out, M=None, C=None, ld=None):
    M = x2d.shape[0] if M is None else M
    C = x2d.shape[1] if C is None else C
    ld = x2d.stride(0) if ld is None else ld
    need = lib().mmfn_colsum_workspace_bytes(M, C)
    _call("mmfn_colsum_f32", ptr(x2d), M, C, ld, ptr(out), ptr(norm_workspace(x2d.device, need)), stream())
    return out


# ---------------------------------------------------------------- pooling / tokens / upsample
def maxpool_fwd(x, y, idx):
    B, H, W, C = x.shape
    _call("mmfn_maxpool3x3s2_fwd_f32", ptr(x), ptr(y), ptr(idx), B, H, W, C, stream())
    return y


def maxpool_bwd(gy, idx, gx):
    B, H, W, C = gx.shape
    _call("mmfn_maxpool3x3s2_bwd_f32", ptr(gy), ptr(idx), ptr(gx), B, H, W, C, stream())
    return gx


def tokens_fwd(feats, pos, vel_w, vel_b, velocity, tok, drop_p=0.0, rng_state=None, rng_stream=0):
    B, S, _, C = feats[0].shape
    arr = _ptr_array(feats)
    _call("mmfn_tokens_fwd_f32", arr, len(feats), B, S, C, ptr(pos), ptr(vel_w), ptr(vel_b), ptr(velocity), ptr(tok),
          float(drop_p), ptr(rng_state), rng_stream, stream())
    return tok


def tokens_bwd(gtok, velocity, dpos, dvel_w, dvel_b, drop_p=0.0, rng_state=None, rng_stream=0):
    B, T, C = gtok.shape
    need = lib().mmfn_tokens_bwd_workspace_bytes(T, C)
    _call("mmfn_tokens_bwd_f32", ptr(gtok), B, T, C, ptr(velocity), ptr(dpos), ptr(dvel_w), ptr(dvel_b), float(drop_p),
          ptr(rng_state), rng_stream, ptr(norm_workspace(gtok.device, need)), stream())


def upsample_add_fwd(feat, tok, out, m):
    B, S, _, C = feat.shape
    T = tok.shape[1]
    _call("mmfn_upsample_add_fwd_f32", ptr(feat), ptr(tok), ptr(out), B, S, C, T, m, stream())
    return out


def upsample_adj(G, gtok, m):
    B, S, _, C = G.shape
    T = gtok.shape[1]
    _call("mmfn_upsample_adj_f32", ptr(G), ptr(gtok), B, S, C, T, m, stream())


def pool_bcast_add(G, gtok, dF, m):
    B, S, _, C = G.shape
    T = gtok.shape[1]
    _call("mmfn_pool_bcast_add_f32", ptr(G), ptr(gtok), ptr(dF), B, S, C, T, m, stream())
    return dF


def gap_sum_fwd(feats, out):
    B, H, W, C = feats[0].shape
    _call("mmfn_gap_sum_fwd_f32", _ptr_array(feats), len(feats), B, H * W, C, ptr(out), stream())
    return out


def gap_sum_bwd(g, outs):
    B, H, W, C = outs[0].shape
    _call("mmfn_gap_sum_bwd_f32", ptr(g), _ptr_array(outs), len(outs), B, H * W, C, stream())


def transpose(inp, out, B, R, Cc):
    _call("mmfn_transpose_f32", ptr(inp), ptr(out), B, R, Cc, stream())
    return out


# ---------------------------------------------------------------- attention
def attention_fwd(q, k, v, ld, o, ldo, lse, B, T, NH, HS, scale, kv_len=None, drop_p=0.0, rng_state=None, rng_stream=0):
    _call("mmfn_attention_fwd_f32", ptr(q), ptr(k), ptr(v), ld, ptr(o), ldo, ptr(lse), B, T, NH, HS, float(scale),
          ptr(kv_len), float(drop_p), ptr(rng_state), rng_stream, stream())
    return o


def attention_bwd(q, k, v, ld, o, dO, ldo, lse, delta, dq, dk, dv, ldg, B, T, NH, HS, scale, kv_len=None, drop_p=0.0,
                  rng_state=None, rng_stream=0):
    _call("mmfn_attention_bwd_f32", ptr(q), ptr(k), ptr(v), ld, ptr(o), ptr(dO), ldo, ptr(lse), ptr(delta), ptr(dq),
          ptr(dk), ptr(dv), ldg, B, T, NH, HS, float(scale), ptr(kv_len), float(drop_p), ptr(rng_state), rng_stream, stream())


# ---------------------------------------------------------------- head / optimizer / ingest / vectornet
def gru_head_fwd(z0, target, w_ih, w_hh, b_ih, b_hh, w_out, b_out, gt, pred, hs, gates, xin, loss_terms, loss, steps):
    B = z0.shape[0]
    _call("mmfn_gru_head_fwd_f32", ptr(z0), ptr(target), ptr(w_ih), ptr(w_hh), ptr(b_ih), ptr(b_hh), ptr(w_out), ptr(b_out),
          ptr(gt), ptr(pred), ptr(hs), ptr(gates), ptr(xin), ptr(loss_terms), ptr(loss), B, steps, stream())


def gru_head_bwd(pred, gt, dpred, gscale, w_ih, w_hh, w_out, hs, gates, xin, dz0, part, steps):
    B = dz0.shape[0]
    _call("mmfn_gru_head_bwd_f32", ptr(pred), ptr(gt), ptr(dpred), float(gscale), ptr(w_ih), ptr(w_hh), ptr(w_out), ptr(hs),
          ptr(gates), ptr(xin), ptr(dz0), ptr(part), B, steps, stream())


def step_advance(step):
    _call("mmfn_step_advance", ptr(step), stream())


def rng_advance(state):
    _call("mmfn_rng_advance", ptr(state), stream())


def adamw(p, g, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, grad_scale=1.0, n=None):
    n = p.numel() if n is None else n
    _call("mmfn_adamw_f32", ptr(p), ptr(g), ptr(m), ptr(v), n, lr, beta1, beta2, eps, weight_decay, ptr(step),
          float(grad_scale), stream())


def ingest_rgb_u8(img_u8, out, crop=256):
    B, H, W, _ = img_u8.shape
    _call("mmfn_ingest_rgb_u8", ptr(img_u8), ptr(out), B, H, W, crop, stream())
    return out


def nchw_to_nhwc(inp, out, mean=None, inv_std=None):
    B, C = inp.shape[0], inp.shape[1]
    P = inp.numel() // (B * C)
    _call("mmfn_nchw_to_nhwc_f32", ptr(inp), ptr(out), B, C, P, ptr(mean), ptr(inv_std), stream())
    return out


def lidar_splat(pts, out, flip_y=False):
    B, N, S = pts.shape
    _call("mmfn_lidar_splat_f32", ptr(pts), B, N, S, ptr(out), 1 if flip_y else 0, stream())
    return out


def lane_to_vector(lane, vec):
    n = lane.shape[-2]
    R = lane.numel() // (n * 5)
    _call("mmfn_lane_to_vector_f32", ptr(lane), ptr(vec), R, n, stream())
    return vec


def polyline_pool_fwd(y, out, arg, R, V, H, last):
    _call("mmfn_polyline_pool_fwd_f32", ptr(y), ptr(out), ptr(arg), R, V, H, 1 if last else 0, stream())
    return out


def polyline_pool_bwd(gout, arg, gy, R, V, H, last):
    _call("mmfn_polyline_pool_bwd_f32", ptr(gout), ptr(arg), ptr(gy), R, V, H, 1 if last else 0, stream())
    return gy


def dropout_apply(inp, out, p, rng_state, rng_stream):
    _call("mmfn_dropout_apply_f32", ptr(inp), ptr(out), inp.numel(), float(p), ptr(rng_state), rng_stream, stream())
    return out


def relu_mask(g, y, out=None):
    out = g if out is None else out
    _call("mmfn_relu_mask_f32", ptr(g), ptr(y), ptr(out), g.numel(), stream())
    return out


# ---------------------------------------------------------------- radar GAT pieces
def elu_fwd(x, y):
    _call("mmfn_elu_fwd_f32", ptr(x), ptr(y), x.numel(), stream())
    return y


def elu_bwd(g, y, dx):
    _call("mmfn_elu_bwd_f32", ptr(g), ptr(y), ptr(dx), g.numel(), stream())
    return dx


def gat_softmax_fwd(e_pre, adj, alpha, p, att, drop_p=0.0, rng_state=None, rng_stream=0):
    R, N = e_pre.shape
    _call("mmfn_gat_softmax_fwd_f32", ptr(e_pre), ptr(adj), float(alpha), ptr(p), ptr(att), R, N, float(drop_p), ptr(rng_state),
          rng_stream, stream())


def gat_softmax_bwd(g_att, p, e_pre, adj, alpha, g_epre, drop_p=0.0, rng_state=None, rng_stream=0):
    R, N = e_pre.shape
    _call("mmfn_gat_softmax_bwd_f32", ptr(g_att), ptr(p), ptr(e_pre), ptr(adj), float(alpha), ptr(g_epre), R, N, float(drop_p),
          ptr(rng_state), rng_stream, stream())


def log_softmax_fwd(x, y, R, C, swap):
    _call("mmfn_log_softmax_fwd_f32", ptr(x), ptr(y), R, C, 1 if swap else 0, stream())
    return y


def log_softmax_bwd(g, y, dx, R, C, swap):
    _call("mmfn_log_softmax_bwd_f32", ptr(g), ptr(y), ptr(dx), R, C, 1 if swap else 0, stream())
    return dx
