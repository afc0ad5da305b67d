"""Thin host wrappers: torch tensors (device memory only) -> C-ABI launches.

Every function enqueues HIP kernels from libmmfn_hip.so on the current torch stream and
returns without synchronising.  Feature maps are NHWC ([B,H,W,C] contiguous); conv weights are
[Cout,KH,KW,Cin] contiguous; Linear weights are [out,in] as in the reference checkpoint.
"""
import ctypes
import json
import os

import torch

from . import _lib
from ._lib import (EPI_KEEP_SLABS, A_COLMAJOR, A_DGRAD, A_IM2COL, A_ROWMAJOR, B_DGRADW, B_IM2COL, B_KN, B_NK,
                   EPI_ACCUM, EPI_BF16_OPERANDS, EPI_BF16X3, EPI_BIAS, EPI_COLSUM_A, EPI_DROPOUT, EPI_GELU, EPI_LN_FOLD, EPI_MASK_AUX, EPI_RELU, EPI_RELU_LAST,
                   EPI_RESIDUAL,
                   GemmDesc, GptBlockDesc, check, lib, ptr, stream)

BF16 = torch.bfloat16   # activations of the bf16 training mode: the wrappers below dispatch on the tensors' dtype
_workspace = {}
_retired = []  # scratch buffers that were outgrown: never freed, captured hipGraphs may still point into them
_lane = 0  # logical execution lane (0 = main stream, 1.. = engine side streams); scratch buffers are per lane


_gemm_dtype = "f32"
# bf16 mode keeps the convolution WEIGHT gradients in fp32: measured on MI355X the bf16 kernel's pixel-contracted form (both
# operands pixel-major, transposed into k-contiguous LDS rows in registers) runs at 28-138 TF/s, slower than the fp32 Winograd-
# domain weight gradient (and than the fp32 direct one at 64 channels), and fp32 weight gradients are the more accurate anyway.
# MMFN_BF16_WGRAD=1 selects the bf16 kernel (kept for its tests and for a transpose-read rewrite).
BF16_WGRAD = False


class precision(object):
    """Context manager: arithmetic of the plain GEMMs launched inside.  "f32" (default): fp32 MFMA, the parity path.
    "bf16": operands rounded to bf16 on their way into LDS, bf16 MFMA, fp32 accumulation / outputs / master weights
    (MMFN_EPI_BF16_OPERANDS; BASELINE configs[2]).  Convolution gather forms always run in fp32."""

    def __init__(self, dtype):
        assert dtype in ("f32", "bf16", "f32x3"), dtype
        self.dtype = dtype

    def __enter__(self):
        global _gemm_dtype
        self.prev, _gemm_dtype = _gemm_dtype, self.dtype

    def __exit__(self, *exc):
        global _gemm_dtype
        _gemm_dtype = self.prev


def current_precision():
    """The arithmetic plain GEMMs launched now would run in (see `precision`)."""
    return _gemm_dtype


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


def current_lane():
    return _lane


class GemmProfiler(object):
    """Brackets every GEMM / convolution of the step with HIP events on the launch stream and tallies
      * its ALGORITHMIC FLOPs: what the operation is (2*M*N*K of the matmul; output pixels x taps x Cin x Cout of the
        convolution, not the zero-inflated implicit-GEMM count) - SURVEY.md section 8d's accounting, and
      * the FLOPs the MFMA units actually EXECUTE for it, which is less when the convolution goes through the Winograd
        transforms (one record then spans filter/input/output transforms + the batched GEMM)."""

    def __init__(self):
        self.records = []   # (start, end, algorithmic flops, tag, algorithmic bytes, executed flops)
        self.suspended = 0

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        return len(self.records), float(sum(r[2] for r in self.records)), ms

    def executed_flops(self):
        return float(sum(r[5] for r in self.records))

    def algo_bytes(self):
        return float(sum(r[4] for r in self.records))

    def by_tag(self):
        torch.cuda.synchronize()
        out = {}
        for s, e, f, tag, _, _ in self.records:
            n, fl, ms = out.get(tag, (0, 0.0, 0.0))
            out[tag] = (n + 1, fl + f, ms + s.elapsed_time(e))
        return out

    def subset(self, prefix):
        """(launches, algorithmic flops, ms) of the records whose tag starts with `prefix` ("bf16 ": the launches that ran on
        the bf16-operand kernel)."""
        torch.cuda.synchronize()
        rows = [r for r in self.records if r[3].startswith(prefix)]
        return len(rows), float(sum(r[2] for r in rows)), sum(r[0].elapsed_time(r[1]) for r in rows)


class _whole_op(object):
    """Profiler bracket around a multi-kernel operation (Winograd convolution): one record with the operation's
    algorithmic FLOPs / bytes; the GEMM launched inside is not recorded a second time."""

    def __init__(self, tag, flops, abytes, executed):
        self.args = (tag, flops, abytes, executed)

    def __enter__(self):
        self.on = _profiler is not None and _profiler.suspended == 0
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        if _profiler is not None:
            _profiler.suspended += 1

    def __exit__(self, *exc):
        if _profiler is not None:
            _profiler.suspended -= 1
        if self.on:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            tag, flops, abytes, executed = self.args
            _profiler.records.append((self.e0, e1, flops, tag, abytes, executed))


_profiler = None


def set_gemm_profiler(p):
    global _profiler
    _profiler = p


# ---------------------------------------------------------------- tile / split-K selection table
# The C library picks (tile, split-K) from an analytic time model; measured on MI355X that choice is 5-20 % off the
# best on a dozen of the step's dominant shapes (tools/gemm_sweep.py).  So the host keeps a table
#   (a_mode, b_mode, M, N, K, batch, conv geometry) -> (tile, splitk)
# loaded from mmfn_amd/tuning/gfx950.json (generated by tools/tune.py on the GPU box) and, when MMFN_AUTOTUNE=1,
# extended at run time: an unknown shape is timed once over the candidate grid on a scratch output (never inside
# graph capture) and the winner cached.  Both fields are plain inputs of the C ABI (mmfn_gemm_desc.tile/.splitk).
_tuned = {}
_TUNE_FILE = os.environ.get("MMFN_TUNING_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "gfx950.json")
AUTOTUNE = os.environ.get("MMFN_AUTOTUNE", "0") == "1"
F32X3 = os.environ.get("MMFN_F32X3", "0") == "1"  # let fp32 GEMMs use the bf16x3 emulation kernel where the table says so
USE_TABLE = os.environ.get("MMFN_TUNING_TABLE", "1") != "0"
_SK_GRID = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96)
# mmfn_gemm_desc.tile -> (rows, columns) of the fp32 kernel's block tile (include/mmfn_hip.h); 7 = 64 x 64 as two waves of 32 x 64
TILE_DIMS = {1: (128, 128), 2: (64, 64), 3: (128, 64), 4: (64, 128), 5: (192, 64), 6: (64, 192), 7: (64, 64)}
TILE_IDS = tuple(int(t) for t in os.environ.get("MMFN_TUNE_TILES", "1,2,3,4,5,6,7").split(","))


def _tune_key(a_mode, b_mode, M, N, K, batch, conv):
    return "%d,%d,%d,%d,%d,%d|%s" % (a_mode, b_mode, M, N, K, max(1, batch), ",".join(str(v) for v in conv) if conv else "")


def load_tuning(path=None):
    path = path or _TUNE_FILE
    if os.path.isfile(path):
        with open(path) as f:
            for k, v in json.load(f).items():
                _tuned[k] = tuple(int(x) for x in v)
    return len(_tuned)


def save_tuning(path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump({k: list(v) for k, v in sorted(_tuned.items())}, f, indent=0)


if USE_TABLE:
    load_tuning()
    if F32X3:  # entries that pick the three-term bf16 emulation kernel for fp32 GEMMs (tools/tune.py under MMFN_F32X3=1)
        load_tuning(os.path.join(os.path.dirname(_TUNE_FILE), "gfx950_f32x3.json"))


def _autotune(d, C, key, reps=3):
    """Time the (tile, split-K) grid for descriptor d on a scratch output; returns the winner or (0, 0) = library default."""
    L = lib()
    M, N, K = d.M, d.N, d.K
    nkt = max(1, K // (32 if d.flags & (EPI_BF16_OPERANDS | EPI_BF16X3) else 16))
    scratch = torch.empty(max(1, d.batch) * M * N + 16, dtype=torch.float32, device=C.device)
    saved = (d.C, d.flags, d.tile, d.splitk, d.workspace, d.strideC, d.ldc)
    # a batched GEMM can split K only if its outputs are packed and the epilogue has no per-batch operand (as in the library)
    batch_split = d.batch <= 1 or (d.strideC == M * N and d.ldc == N and not d.flags & (EPI_RESIDUAL | EPI_MASK_AUX | EPI_ACCUM))
    d.C = ptr(scratch)
    d.flags = d.flags & ~EPI_ACCUM
    if d.batch > 1:
        d.strideC = M * N
    d.ldc = N

    def run(tile, sk):
        d.tile, d.splitk = tile, sk
        need = L.mmfn_gemm_workspace_bytes(ctypes.byref(d))
        d.workspace = ptr(workspace(need, C.device)) if need > 0 else None
        st = stream()
        for _ in range(2):
            if L.mmfn_gemm_f32(ctypes.byref(d), st) != 0:
                return None
        best = None
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                L.mmfn_gemm_f32(ctypes.byref(d), st)
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1) / 4
            best = t if best is None else min(best, t)
        return best

    base = run(0, 0)
    results = []
    plain = d.a_mode in (A_ROWMAJOR, A_COLMAJOR) and d.b_mode in (B_NK, B_KN)
    modes = [0]
    # (not with MMFN_EPI_COLSUM_A: the column sums are formed from the fp32 operand fragments of the native kernel only)
    if F32X3 and plain and not d.flags & (EPI_BF16_OPERANDS | EPI_BF16X3 | EPI_COLSUM_A) and K % 32 == 0:
        modes.append(EPI_BF16X3)  # fp32 on the bf16 pipe (three-term split) competes with the native fp32 MFMA kernel
    if base is not None:
        for mode in modes:
            d.flags = (saved[1] & ~EPI_ACCUM) | mode
            emu = bool(d.flags & (EPI_BF16_OPERANDS | EPI_BF16X3))
            kt = max(1, K // (32 if emu else 16))
            for tile in ((1, 2) if emu else TILE_IDS):
                bm, bn = TILE_DIMS[tile]
                blocks = -(-M // bm) * -(-N // bn)
                for sk in _SK_GRID:
                    if sk > 1 and (not batch_split or sk > kt // 4):
                        break
                    if blocks * sk * max(1, d.batch) < 96 or blocks * sk > 16384:
                        continue
                    t = run(tile, sk)
                    if t is not None:
                        results.append((t, tile, sk, mode))
    d.C, d.flags, d.tile, d.splitk, d.workspace, d.strideC, d.ldc = saved
    choice = (0, 0)
    if results:
        t, tile, sk, mode = min(results)
        if t < 0.97 * base:
            choice = (tile, sk, mode) if mode else (tile, sk)
    _tuned[key] = choice
    return choice


def workspace(nbytes, device):
    """Grow-only scratch buffer (split-K slabs); never freed so captured graphs stay valid.
    One buffer per (device, lane): branches running concurrently on side streams must not share slabs."""
    key = (str(device), _lane)
    buf = _workspace.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("split-K workspace must be sized before graph capture")
        if buf is not None:
            _retired.append(buf)
        buf = torch.empty(max(nbytes // 4 + 1, 1 << 22), dtype=torch.float32, device=device)
        _workspace[key] = buf
    return buf


def _f32c(t, name):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda, name
    return t


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_mode=A_ROWMAJOR, b_mode=B_NK, bias=None, res=None, ldr=0,
         aux=None, ldaux=0, relu=False, gelu=False, accum=False, relu_last=False, drop_p=0.0, rng_state=None, rng_stream=0,
         conv=None, splitk=0, tile=0, batch=1, strideA=0, strideB=0, strideC=0, ln_fold=None, colsum=None, slab_info=None):
    """colsum ([M] fp32, TN form): also the column sums of the A operand (MMFN_EPI_COLSUM_A: a Linear's bias gradient from its
    weight-gradient GEMM).
    ln_fold = (c1 [N], mean [M] or None, rstd [M] or None, eps): C = LayerNorm(A) . B^T + bias with B / bias the folded operands
    of ln_fold_weights (MMFN_EPI_LN_FOLD, include/mmfn_hip.h); mean / rstd receive the row statistics."""
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
    if relu_last:
        flags |= EPI_RELU_LAST
    if ln_fold is not None:
        c1, ln_mean, ln_rstd, ln_eps = ln_fold
        assert bias is not None and a_mode == A_ROWMAJOR and b_mode == B_NK and batch <= 1
        flags |= EPI_LN_FOLD
        d.ln_c1, d.ln_mean, d.ln_rstd, d.ln_eps = ptr(c1), ptr(ln_mean), ptr(ln_rstd), float(ln_eps)
        splitk = 1
    # bf16 mode: plain GEMMs and - as DIRECT convolutions - the im2col forward / flipped data gradient and the weight gradient
    # (the library falls back to the fp32 kernel for what the bf16 kernel does not cover: 7x7 stems, stride-2 data gradients)
    bf16 = _gemm_dtype != "f32" and (conv is None or (_gemm_dtype == "bf16" and (
        (a_mode, b_mode) == (A_IM2COL, B_NK) or (BF16_WGRAD and (a_mode, b_mode) == (A_COLMAJOR, B_IM2COL)))))
    if bf16:
        flags |= EPI_BF16X3 if _gemm_dtype == "f32x3" else EPI_BF16_OPERANDS
    if colsum is not None:
        assert (a_mode, b_mode) == (A_COLMAJOR, B_KN) and batch <= 1 and not bf16
        flags |= EPI_COLSUM_A
        d.colsum = ptr(colsum)
    d.flags = flags
    d.splitk = splitk
    d.tile = tile
    L = lib()
    if tile == 0 and (splitk == 0 or ln_fold is not None) and (USE_TABLE or AUTOTUNE):
        key = ((_gemm_dtype + "|") if bf16 else "") + _tune_key(a_mode, b_mode, M, N, K, batch, conv)
        cfg = _tuned.get(key)
        if cfg is None and AUTOTUNE and _profiler is None and not torch.cuda.is_current_stream_capturing() and ln_fold is None:
            cfg = _autotune(d, C, key)
        if cfg is not None and len(cfg) == 3 and colsum is not None:
            # a table entry that moves this shape to the three-term bf16 emulation kernel (MMFN_F32X3=1): that kernel cannot also
            # return the A operand's column sums (mmfn_gemm_f32 answers MMFN_EINVAL to COLSUM_A + BF16X3), and the entry's
            # (tile, split) were measured for it - the fused bias gradient keeps the native kernel at the library's default
            cfg = None
        if cfg is not None and (len(cfg) == 2 or F32X3):
            d.tile, d.splitk = cfg[0], cfg[1]
            if len(cfg) == 3 and ln_fold is None:
                d.flags |= cfg[2]
        elif cfg is not None:  # table entry wants the emulation kernel but it is switched off: library default
            pass
    if ln_fold is not None:   # the plain GEMM's tile of the same shape (table entry or the library's model); never split - a tile
        d.splitk = 1          # that does not divide M x N falls back to 64 x 64 inside the launcher
    # slab_info (a dict): a launch that splits K leaves its slices un-combined (MMFN_EPI_KEEP_SLABS) and reports them here as
    # slab_info["splits"] (> 1) and slab_info["slabs"] (the workspace tensor, [splits][batch][M][N]); with splits == 1 C holds the result
    keep = slab_info is not None and flags == 0 and not bf16 and not (d.flags & (EPI_BF16X3 | EPI_BF16_OPERANDS))
    if keep:
        d.flags |= EPI_KEEP_SLABS
    need = L.mmfn_gemm_workspace_bytes(ctypes.byref(d))
    wsbuf = None
    if need > 0:
        wsbuf = workspace(need, C.device)
        d.workspace = ptr(wsbuf)
    if slab_info is not None:
        slab_info["splits"] = L.mmfn_gemm_f32_splits(ctypes.byref(d)) if (keep and wsbuf is not None) else 1
        slab_info["slabs"] = wsbuf
        if slab_info["splits"] <= 1:
            d.flags &= ~EPI_KEEP_SLABS
    if _profiler is None or _profiler.suspended:
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
    _profiler.records.append((e0, e1, flops, ("bf16 " if bf16 else "") + tag, abytes, flops))
    return C


# ---------------------------------------------------------------- Linear
def linear_fwd(x, w, bias=None, out=None, **epi):
    """out[M,N] = x[M,K] @ w[N,K]^T (+bias, epilogue)."""
    M, K = x.shape
    N = w.shape[0]
    if x.dtype == BF16:   # bf16 mode: w is the bf16 shadow of the weight
        from . import ops16
        return ops16.linear_fwd(x, w, bias, out=out, **epi)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    return gemm(x, w, out, M, N, K, x.stride(0), w.stride(0), out.stride(0), A_ROWMAJOR, B_NK, bias=bias, **epi)


def linear_dx(dy, w, out=None, **epi):
    """dx[M,K] = dy[M,N] @ w[N,K].  bf16 mode: w is the TRANSPOSED bf16 shadow [K,N]."""
    if dy.dtype == BF16:
        from . import ops16
        return ops16.linear_dx(dy, w, out=out, **epi)
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty(M, K, dtype=torch.float32, device=dy.device)
    return gemm(dy, w, out, M, K, N, dy.stride(0), w.stride(0), out.stride(0), A_ROWMAJOR, B_KN, **epi)


FUSE_COLSUM = os.environ.get("MMFN_FUSE_COLSUM", "1") != "0"   # bias gradients from the weight-gradient GEMM (MMFN_EPI_COLSUM_A)


def linear_dw(dy, x, out=None, db=None, **epi):
    """dw[N,K] = dy[M,N]^T @ x[M,K];  db ([N], optional) = the column sums of dy = the bias gradient: formed inside the same GEMM
    where the fast TN kernel runs (fp32, 16-byte aligned operands, M a multiple of 16, N and K multiples of 4), else by colsum()."""
    if dy.dtype == BF16:
        from . import ops16
        assert db is None
        return ops16.linear_dw(dy, x, out, **epi)
    M, N = dy.shape
    K = x.shape[1]
    if out is None:
        out = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    if db is not None:
        fusable = (FUSE_COLSUM and _gemm_dtype == "f32" and M % 16 == 0 and N % 4 == 0 and K % 4 == 0 and N >= 4 and K >= 4
                   and dy.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)
        if not fusable:
            colsum(dy, db)
            db = None
    return gemm(dy, x, out, N, K, M, dy.stride(0), x.stride(0), out.stride(0), A_COLMAJOR, B_KN, colsum=db, **epi)


# ---------------------------------------------------------------- Convolution (NHWC)
def conv_geom(x_shape, w_shape, stride, pad):
    B, H, W, Cin = x_shape
    Cout, KH, KW, Cin2 = w_shape
    assert Cin == Cin2
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    return (H, W, Cin, OH, OW, Cout, KH, KW, stride, pad), (B, OH, OW, Cout)


_wino = {}
WINOGRAD_MIN_CHANNELS = int(os.environ.get("MMFN_WINOGRAD_MIN_C", "64"))  # 0 disables the Winograd path


def _wino_scratch(n, device):
    """Per-(device, lane) scratch for the transformed filter / activations / products (grow-only, sized before capture)."""
    key = (str(device), _lane)
    buf = _wino.get(key)
    if buf is None or buf.numel() < n:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("Winograd scratch must be sized before graph capture")
        if buf is not None:
            _retired.append(buf)
        buf = torch.empty(n, dtype=torch.float32, device=device)
        _wino[key] = buf
    return buf


def winograd_ok(x_shape, w_shape, stride, pad, epi, for_wgrad=False):
    Co, KH, KW, Ci = w_shape
    B, H, W, _ = x_shape
    if _gemm_dtype == "bf16" and not for_wgrad:
        # bf16 mode runs every 3x3 convolution as a direct implicit GEMM on the bf16 MFMA pipe: rounding Winograd-domain
        # operands to bf16 amplifies the error (transform gains up to ~100), and at 16x the fp32 MFMA rate the 4x FLOP saving
        # no longer pays for the 2.25x larger transformed tensors
        return False
    return (WINOGRAD_MIN_CHANNELS > 0 and KH == 3 and KW == 3 and stride == 1 and pad == 1 and H % 2 == 0 and W % 2 == 0
            and min(Ci, Co) >= WINOGRAD_MIN_CHANNELS and Ci % 16 == 0 and Co % 16 == 0 and set(epi) <= {"res", "ldr"})


WINOGRAD_F4 = True   # F(4x4,3x3) wherever the image sides are multiples of 4 (tests flip it to reach the F(2x2) kernels)


def winograd_v_numel(x_shape):
    """Elements of the transformed-input tensor V of the F(4x4,3x3) path for an NHWC input of this shape."""
    B, H, W, C = x_shape
    return 36 * B * (H // 4) * (W // 4) * C


def winograd_f4_ok(x_shape, w_shape, stride, pad):
    """The F(4x4,3x3) form of the forward (what the fused BatchNorm-apply input transform exists for)."""
    return winograd_ok(x_shape, w_shape, stride, pad, {}) and WINOGRAD_F4 and x_shape[1] % 4 == 0 and x_shape[2] % 4 == 0


def wino_weight_group(table, n_layers, total):
    """All filter transforms of the step in one launch (table: uint8 device tensor of 32-byte records, see the header)."""
    _call("mmfn_wino_weight_group_f32", ptr(table), n_layers, total, stream())


def make_wino_group_table(pairs, device):
    """pairs: [(w [Co,3,3,Ci] storage tensor, U flat tensor)] -> (device table, n, total)."""
    import struct
    blob, start = b"", 0
    for w, u in pairs:
        Co, _, _, Ci = w.shape
        blob += struct.pack("<QQiiq", w.data_ptr(), u.data_ptr(), Co, Ci, start)
        start += Co * Ci
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    return t, len(pairs), start


def conv2d_winograd(x, w, out, res=None, m=None, keep_v=None, bn=None, keep_u=None, u_ready=False, x_bn=None):
    """3x3 stride-1 'same' convolution as Winograd F(m x m, 3x3): streaming transforms around one (m+2)^2-batch GEMM.
    m = 4 when the image size allows (4x fewer MFMA FLOPs), else 2.  keep_v (F(4x4) only): a per-layer buffer that
    receives the transformed input, so the weight gradient can reuse it instead of transforming x again.
    x_bn = (res or None, mean, rstd, weight, bias, relu, y_out or None) (F(4x4) only): x is the PRODUCER's convolution output and
    the convolution's input is [relu](bn(x) [+ res]), applied inside the input transform (mmfn_wino_input_bn_f32); y_out also
    receives that activation as a tensor."""
    B, H, W, Ci = x.shape
    Co = w.shape[0]
    if m is None:
        m = 4 if (WINOGRAD_F4 and H % 4 == 0 and W % 4 == 0) else 2
    n = (m + 2) * (m + 2)
    T = B * (H // m) * (W // m)
    buf = _wino_scratch(n * (Co * Ci + T * Ci + T * Co), x.device)
    U = buf[:n * Co * Ci]
    V = buf[n * Co * Ci:n * (Co * Ci + T * Ci)]
    if keep_v is not None and m == 4:
        V = keep_v.view(-1)[:n * T * Ci]
    if keep_u is not None and m == 4:   # the transformed filter, kept for the adjoint data gradient (conv2d_bwd_winograd)
        U = keep_u.view(-1)[:n * Co * Ci]
    Mt = buf[n * (Co * Ci + T * Ci):n * (Co * Ci + T * Ci + T * Co)]
    st = stream()
    with _whole_op("conv %dx%d c%d->%d k3 s1 (winograd F%d)" % (H, W, Ci, Co, m), 2.0 * B * H * W * Co * 9 * Ci,
                   4.0 * (B * H * W * (Ci + Co) + 9 * Co * Ci), 2.0 * n * T * Co * Ci):
        if not (u_ready and keep_u is not None and m == 4):   # u_ready: this layer's U was written by wino_weight_group
            _call("mmfn_wino_weight_f32", ptr(w), ptr(U), Co, Ci, m, st)
        if x_bn is not None:
            assert m == 4
            xres, xmean, xrstd, xw, xb, xrelu, y_out = x_bn
            _call("mmfn_wino_input_bn_f32", ptr(x), ptr(xres), ptr(xmean), ptr(xrstd), ptr(xw), ptr(xb), 1 if xrelu else 0, ptr(y_out),
                  ptr(V), B, H, W, Ci, st)
        else:
            _call("mmfn_wino_input_f32", ptr(x), ptr(V), B, H, W, Ci, m, st)
        gemm(V, U, Mt, T, Co, Ci, Ci, Ci, Co, A_ROWMAJOR, B_NK, batch=n, strideA=T * Ci, strideB=Co * Ci, strideC=T * Co)
        if bn is not None and m == 4 and res is None and 256 % (Co // 4) == 0:
            # BatchNorm batch statistics come out of the output transform; only the tiny finalize kernel remains
            nblk = ctypes.c_int(0)
            ws = norm_workspace(x.device)
            _call("mmfn_wino_output_stats_f32", ptr(Mt), ptr(out), ptr(ws), ctypes.cast(ctypes.pointer(nblk), ctypes.c_void_p), B, H, W, Co, st)
            mean, rstd, rm, rv, nbt, eps, momentum = bn
            _call("mmfn_bn_finalize_stats_f32", ptr(ws), nblk.value, B * H * W, Co, eps, momentum, ptr(mean), ptr(rstd), ptr(rm),
                  ptr(rv), ptr(nbt), st)
            return out, True
        _call("mmfn_wino_output_f32", ptr(Mt), ptr(res), ptr(out), B, H, W, Co, m, st)
    return (out, False) if bn is not None else out


def conv2d_fwd_bn_stats(x, w, stride, pad, out, mean, rstd, running_mean, running_var, nbt, eps, momentum, keep_v=None, keep_u=None,
                        u_ready=False, x_bn=None):
    """Training-mode conv followed by the BatchNorm batch statistics of its output (model_vec.py:509-593: every conv of the
    trunks is followed by a BatchNorm2d).  On the Winograd path the statistics are a by-product of the output transform.
    x_bn: see conv2d_winograd (the caller checked winograd_f4_ok)."""
    if winograd_ok(x.shape, w.shape, stride, pad, {}):
        _, fused = conv2d_winograd(x, w, out, keep_v=keep_v, keep_u=keep_u, u_ready=u_ready, x_bn=x_bn,
                                   bn=(mean, rstd, running_mean, running_var, nbt, eps, momentum))
        if fused:
            return out
    else:
        assert x_bn is None
        conv2d_fwd(x, w, stride, pad, out=out, keep_v=keep_v, keep_u=keep_u, u_ready=u_ready)
    M = out.numel() // out.shape[-1]
    bn_train_stats(out.view(M, out.shape[-1]), mean, rstd, running_mean, running_var, nbt, eps, momentum)
    return out


def conv2d_fwd(x, w, stride, pad, out=None, keep_v=None, keep_u=None, u_ready=False, x_bn=None, **epi):
    """y[B,OH,OW,Cout] = conv(x[B,H,W,Cin], w[Cout,KH,KW,Cin]).  x_bn: see conv2d_winograd (the caller checked winograd_f4_ok)."""
    g, oshape = conv_geom(x.shape, w.shape, stride, pad)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    if winograd_ok(x.shape, w.shape, stride, pad, epi):
        assert epi.get("ldr", oshape[3]) == oshape[3]
        return conv2d_winograd(x, w, out, res=epi.get("res"), keep_v=keep_v, keep_u=keep_u, u_ready=u_ready, x_bn=x_bn)
    assert x_bn is None
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
        if buf is not None:
            _retired.append(buf)
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


WINOGRAD_WGRAD = True


def winograd_wgrad_ok(x_shape, w_shape, stride, pad):
    # also in bf16 mode (where the forward is a direct convolution): the weight gradient then runs in the Winograd domain in
    # fp32, transforming x itself
    return (WINOGRAD_WGRAD and WINOGRAD_F4 and winograd_ok(x_shape, w_shape, stride, pad, {}, for_wgrad=True)
            and x_shape[1] % 4 == 0 and x_shape[2] % 4 == 0)


def conv2d_wgrad_winograd(dy, x, out, v=None):
    """Weight gradient of a 3x3 stride-1 'same' convolution in the F(4x4,3x3) domain: 36 [Co x tiles] x [tiles x Ci] GEMMs.
    v: the transformed input kept by the forward (conv2d_fwd(keep_v=...)); recomputed from x when absent."""
    B, H, W, Ci = x.shape
    Co = dy.shape[3]
    T = B * (H // 4) * (W // 4)
    buf = _wino_scratch(36 * (Co * Ci + T * Ci + T * Co), x.device)
    dU = buf[:36 * Co * Ci]
    V = buf[36 * Co * Ci:36 * (Co * Ci + T * Ci)]
    dMt = buf[36 * (Co * Ci + T * Ci):36 * (Co * Ci + T * Ci + T * Co)]
    st = stream()
    with _whole_op("wgrad %dx%d c%d->%d k3 s1 (winograd F4)" % (H, W, Ci, Co), 2.0 * B * H * W * Co * 9 * Ci,
                   4.0 * (B * H * W * (Ci + Co) + 9 * Co * Ci), 2.0 * 36 * T * Co * Ci), precision("f32"):
        _wgrad_winograd_body(dy, x, out, v, B, H, W, Ci, Co, T, dU, V, dMt, st)   # never rounded in the Winograd domain
    return out


def _wgrad_winograd_body(dy, x, out, v, B, H, W, Ci, Co, T, dU, V, dMt, st, bn=None):
    if v is not None:
        V = v.view(-1)[:36 * T * Ci]
    else:
        _call("mmfn_wino_input_f32", ptr(x), ptr(V), B, H, W, Ci, 4, st)
    if bn is not None:
        g, ymask, co, mean, rstd, weight, relu_bias, means, ge_out = bn
        _call("mmfn_wino_outgrad_bn_f32", ptr(g), ptr(ymask), ptr(co), ptr(mean), ptr(rstd), ptr(weight), ptr(relu_bias), ptr(means),
              ptr(ge_out), ptr(dMt), B, H, W, Co, st)
    else:
        _call("mmfn_wino_outgrad_f32", ptr(dy), ptr(dMt), B, H, W, Co, st)
    # the split-K combine of this GEMM (K = tiles: thousands) runs inside the output transform, which sums the slices as it reads
    info = {} if (WGRAD_SLABS and Ci % 64 == 0) else None
    gemm(dMt, V, dU, Co, Ci, T, Co, Ci, Ci, A_COLMAJOR, B_KN, batch=36, strideA=T * Co, strideB=T * Ci, strideC=Co * Ci, slab_info=info)
    if info is not None and info["splits"] > 1:
        _call("mmfn_wino_wgrad_out_slabs_f32", ptr(info["slabs"]), info["splits"], ptr(out), Co, Ci, st)
    else:
        _call("mmfn_wino_wgrad_out_f32", ptr(dU), ptr(out), Co, Ci, st)
    return out


WGRAD_SLABS = True   # False: a split-K combine launch per Winograd weight gradient (the test of the slab path compares both)
WINOGRAD_ADJOINT_DGRAD = True


def winograd_adjoint_ok(x_shape, w_shape, stride, pad):
    """The adjoint data gradient covers whatever the F(4x4,3x3) weight gradient covers (image sides multiples of 4)."""
    return WINOGRAD_ADJOINT_DGRAD and winograd_wgrad_ok(x_shape, w_shape, stride, pad)


def conv2d_bwd_winograd(dy, x, u, dw_out, dx_out, v=None, res=None, bn=None):
    """Weight AND data gradient of a 3x3 stride-1 'same' convolution in the F(4x4,3x3) domain, sharing the transformed
    output gradient dM = A dy A^T:
        dw = G^T [ sum_tiles dM^T . V ] G          (V = B^T x B, kept by the forward or recomputed)
        dx = overlap-add( B (dM . U) B^T ) (+ res)  (U = G w G^T, kept by the forward: the ADJOINT of the forward pipeline,
                                                     no flipped filter, no second filter / input transform)
    bn = (g, ymask or None, conv_out, mean, rstd, weight, relu_bias or None, means, ge_out or None), all NHWC / [C]: dy is then the
    BatchNorm backward of g (reductions already done by bn_bwd_reduce) formed on the fly inside the transform; the `dy` argument
    only gives the shape.  relu_bias (with ymask None): the ReLU mask is recomputed from conv_out (bn_bwd).
    x: the convolution's input, or - with v given - anything carrying its .shape."""
    B, H, W, Ci = x.shape
    Co = dy.shape[3]
    T = B * (H // 4) * (W // 4)
    buf = _wino_scratch(36 * (Co * Ci + T * Ci + T * Co), x.device)
    dU = buf[:36 * Co * Ci]
    Vs = buf[36 * Co * Ci:36 * (Co * Ci + T * Ci)]
    dMt = buf[36 * (Co * Ci + T * Ci):36 * (Co * Ci + T * Ci + T * Co)]
    st = stream()
    conv_flops = 2.0 * B * H * W * Co * 9 * Ci
    conv_bytes = 4.0 * (B * H * W * (Ci + Co) + 9 * Co * Ci)
    with _whole_op("wgrad %dx%d c%d->%d k3 s1 (winograd F4)" % (H, W, Ci, Co), conv_flops, conv_bytes, 2.0 * 36 * T * Co * Ci):
        _wgrad_winograd_body(dy, x, dw_out, v, B, H, W, Ci, Co, T, dU, Vs, dMt, st, bn=bn)
    with _whole_op("dgrad %dx%d c%d->%d k3 s1 (winograd F4 adjoint)" % (H, W, Ci, Co), conv_flops, conv_bytes, 2.0 * 36 * T * Co * Ci):
        dV = Vs  # the scratch V region is free again: with a kept V it was never used, otherwise the wgrad GEMM is done with it
        gemm(dMt, u.view(-1)[:36 * Co * Ci], dV, T, Ci, Co, Co, Ci, Ci, A_ROWMAJOR, B_KN, batch=36, strideA=T * Co, strideB=Co * Ci,
             strideC=T * Ci)
        _call("mmfn_wino_input_adjoint_f32", ptr(dV), ptr(res), ptr(dx_out), B, H, W, Ci, st)
    return dw_out, dx_out


def conv2d_wgrad(dy, x, w_shape, stride, pad, out=None, v=None, **epi):
    """dw[Cout,KH,KW,Cin] = sum over pixels of dy (x) im2col(x)."""
    g, oshape = conv_geom(x.shape, w_shape, stride, pad)
    assert tuple(dy.shape) == oshape
    if out is None:
        out = torch.empty(w_shape, dtype=torch.float32, device=dy.device)
    if not epi and winograd_wgrad_ok(x.shape, w_shape, stride, pad):
        return conv2d_wgrad_winograd(dy, x, out, v=v)
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
        if buf is not None:
            _retired.append(buf)
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
    _call("mmfn_bn_train_stats_bf16" if x2d.dtype == BF16 else "mmfn_bn_train_stats_f32", ptr(x2d), M, C, eps, momentum, ptr(mean),
          ptr(rstd), ptr(running_mean), ptr(running_var), ptr(nbt), ptr(norm_workspace(x2d.device)), stream())


def bn_finalize_stats(partials, nblk, M, C, mean, rstd, running_mean, running_var, nbt, eps=1e-5, momentum=0.1):
    """Second half of bn_train_stats for producers that emit the per-block (sum, sum of squares) rows themselves (the bf16
    convolution's epilogue, ops16.conv2d_fwd(stats=...)): partials [nblk, 2, C] fp64."""
    _call("mmfn_bn_finalize_stats_f32", ptr(partials), nblk, M, C, eps, momentum, ptr(mean), ptr(rstd), ptr(running_mean),
          ptr(running_var), ptr(nbt), stream())


def bn_fold(w, gamma, beta, running_mean, running_var, eps, w_out, b_out):
    """Eval-mode BatchNorm folded into the convolution before it: w_out = w * s per output channel, b_out = beta - mean * s."""
    Cout = w.shape[0]
    _call("mmfn_bn_fold_f32", ptr(w), Cout, w.numel() // Cout, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps),
          ptr(w_out), ptr(b_out), stream())


def bn_eval_prepare(running_mean, running_var, mean, rstd, eps=1e-5):
    _call("mmfn_bn_eval_prepare_f32", ptr(running_mean), ptr(running_var), eps, mean.numel(), ptr(mean), ptr(rstd), stream())


def bn_apply(x2d, y2d, mean, rstd, weight, bias, relu, res=None):
    M, C = x2d.shape
    if y2d.dtype == BF16:   # x: the convolution output, bf16 or (stems) fp32
        _call("mmfn_bn_apply_bf16", ptr(x2d), 0 if x2d.dtype == BF16 else 1, ptr(res), ptr(y2d), M, C, ptr(mean), ptr(rstd),
              ptr(weight), ptr(bias), 1 if relu else 0, stream())
        return y2d
    _call("mmfn_bn_apply_f32", ptr(x2d), ptr(res), ptr(y2d), M, C, ptr(mean), ptr(rstd), ptr(weight), ptr(bias),
          1 if relu else 0, stream())
    return y2d


def bn_bwd(g2d, y2d, x2d, mean, rstd, weight, dx, dweight, dbias, ge_out=None, relu_bias=None):
    """relu_bias (the BatchNorm's bias) with y2d None: the ReLU mask is recomputed from x2d - the forward applied this BatchNorm
    inside its consumer's Winograd input transform and never wrote y (conv2d_winograd(x_bn=...))."""
    M, C = x2d.shape
    if g2d.dtype != BF16 and (x2d.dtype == BF16 or (y2d is not None and y2d.dtype == BF16)):
        raise TypeError("bn_bwd: fp32 output gradient with bf16 activations (every activation-side tensor of the bf16 mode is bf16)")
    if g2d.dtype == BF16:   # x / dx: bf16, or both fp32 (stems)
        assert dx.dtype == x2d.dtype
        _call("mmfn_bn_bwd_bf16", ptr(g2d), ptr(y2d), ptr(x2d), 0 if x2d.dtype == BF16 else 1, M, C, ptr(mean), ptr(rstd), ptr(weight),
              ptr(dx), ptr(ge_out), ptr(dweight), ptr(dbias), ptr(norm_workspace(x2d.device)), stream())
        return dx
    _call("mmfn_bn_bwd_f32", ptr(g2d), ptr(y2d), ptr(x2d), M, C, ptr(mean), ptr(rstd), ptr(weight), ptr(relu_bias), ptr(dx),
          ptr(ge_out), ptr(dweight), ptr(dbias), ptr(norm_workspace(x2d.device)), stream())
    return dx


def bn_bwd_reduce(g2d, y2d, x2d, mean, rstd, dweight, dbias, means, relu_wb=None):
    """The reductions of bn_bwd only (dweight, dbias, means[2][C]); conv2d_bwd_winograd(bn=...) applies them inside its
    output-gradient transform.  relu_wb = (weight, bias) with y2d None: the recomputed ReLU mask (bn_bwd)."""
    M, C = x2d.shape
    rw, rb = relu_wb if relu_wb is not None else (None, None)
    _call("mmfn_bn_bwd_reduce_f32", ptr(g2d), ptr(y2d), ptr(x2d), M, C, ptr(mean), ptr(rstd), ptr(rw), ptr(rb), ptr(dweight),
          ptr(dbias), ptr(means), ptr(norm_workspace(x2d.device)), stream())
    return means


ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


def layernorm_fwd(x, w, b, y, mean, rstd, act=ACT_NONE, eps=1e-5):
    """y bf16 (the bf16 mode): x bf16, or fp32 - the transformers' residual stream, which stays fp32 in that mode."""
    M, C = x.shape
    if y.dtype == BF16:
        _call("mmfn_layernorm_fwd_bf16", ptr(x), 0 if x.dtype == BF16 else 1, ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), M, C, eps, act,
              stream())
        return y
    _call("mmfn_layernorm_fwd_f32", ptr(x), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), M, C, eps, act, stream())
    return y


def layernorm_bwd(g, x, w, b, mean, rstd, dx, dw, db, act=ACT_NONE, dres=None, dx_dropped=None, drop_p=0.0, rng_state=None,
                  rng_stream=0, dx_colsum=None):
    """dx_dropped: optional second output dx * dropout keep-scale (the mask of the residual branch that consumes dx).
    dx_colsum: optional [C] output, column sums of dx_dropped (of dx without a dropped copy): the next bias gradient."""
    M, C = x.shape
    if g.dtype == BF16:   # the two halves separately (same kernels as the fused entry)
        rows = layernorm_bwd_rows(M)
        part = norm_workspace(x.device).view(torch.float32)[:rows * 3 * C].view(rows, 3, C)
        layernorm_bwd_partial(g, x, w, b, mean, rstd, dx, part, act, dres=dres, dx_dropped=dx_dropped, drop_p=drop_p,
                              rng_state=rng_state, rng_stream=rng_stream, want_colsum=dx_colsum is not None)
        layernorm_bwd_finalize(part, rows, C, dw, db, dx_colsum)
        return dx
    _call("mmfn_layernorm_bwd_drop_f32", ptr(g), ptr(x), ptr(w), ptr(b), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dw),
          ptr(db), M, C, act, ptr(dx_dropped), float(drop_p), ptr(rng_state), rng_stream, ptr(dx_colsum),
          ptr(norm_workspace(x.device)), stream())
    return dx


def layernorm_bwd_rows(M):
    return lib().mmfn_layernorm_bwd_rows(M)


def layernorm_bwd_partial(g, x, w, b, mean, rstd, dx, partials, act=ACT_NONE, dres=None, dx_dropped=None, drop_p=0.0,
                          rng_state=None, rng_stream=0, want_colsum=False):
    """First half of layernorm_bwd: dx (and dx_dropped) now, the row reductions as partial rows in `partials`
    ([layernorm_bwd_rows(M), 3 or 2, C] floats) for layernorm_bwd_finalize - which may run later, on another stream."""
    M, C = x.shape
    if g.dtype == BF16:
        # bf16 mode: g and the dropped copy are GEMM operands (bf16); x / dres / dx are bf16 too, or - inside the fusion transformers -
        # the fp32 residual stream and its gradient
        f32_stream = x.dtype == torch.float32
        assert dx.dtype == x.dtype and (dres is None or dres.dtype == x.dtype) and (dx_dropped is None or dx_dropped.dtype == BF16)
        _call("mmfn_layernorm_bwd_partial_bf16", ptr(g), ptr(x), 1 if f32_stream else 0, ptr(w), ptr(b), ptr(mean), ptr(rstd), ptr(dres),
              ptr(dx), M, C, act, ptr(dx_dropped), float(drop_p), ptr(rng_state), rng_stream, 1 if want_colsum else 0, ptr(partials),
              stream())
        return dx
    _call("mmfn_layernorm_bwd_partial_f32", ptr(g), ptr(x), ptr(w), ptr(b), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), M, C, act,
          ptr(dx_dropped), float(drop_p), ptr(rng_state), rng_stream, 1 if want_colsum else 0, ptr(partials), stream())
    return dx


def layernorm_bwd_finalize(partials, rows, C, dw, db, dx_colsum=None):
    _call("mmfn_layernorm_bwd_finalize_f32", ptr(partials), rows, C, ptr(dw), ptr(db), ptr(dx_colsum), stream())


def layernorm_finalize_table(entries, device):
    """Device pointer table for layernorm_bwd_finalize_batched: entries = [(partials, dw, db, dx_colsum or None), ...]."""
    flat = []
    for part, dw, db, cs in entries:
        flat += [part.data_ptr(), dw.data_ptr(), db.data_ptr(), 0 if cs is None else cs.data_ptr()]
    return torch.tensor(flat, dtype=torch.int64).to(device)


def layernorm_bwd_finalize_batched(table, n, rows, C):
    _call("mmfn_layernorm_bwd_finalize_batched_f32", ptr(table), n, rows, C, stream())


def colsum(x2d, out, M=None, C=None, ld=None):
    M = x2d.shape[0] if M is None else M
    C = x2d.shape[1] if C is None else C
    ld = x2d.stride(0) if ld is None else ld
    need = lib().mmfn_colsum_workspace_bytes(M, C)
    _call("mmfn_colsum_bf16" if x2d.dtype == BF16 else "mmfn_colsum_f32", ptr(x2d), M, C, ld, ptr(out),
          ptr(norm_workspace(x2d.device, need)), stream())
    return out


def maxpool_fwd(x, y, idx):
    B, H, W, C = x.shape
    _call("mmfn_maxpool3x3s2_fwd_bf16" if x.dtype == BF16 else "mmfn_maxpool3x3s2_fwd_f32", ptr(x), ptr(y), ptr(idx), B, H, W, C, stream())
    return y


def maxpool_bwd(gy, idx, gx):
    B, H, W, C = gx.shape
    _call("mmfn_maxpool3x3s2_bwd_bf16" if gy.dtype == BF16 else "mmfn_maxpool3x3s2_bwd_f32", ptr(gy), ptr(idx), ptr(gx), B, H, W, C, stream())
    return gx


def _frames(frames, n):
    """Host int32[n] of frames per sample for each modality (None: one each - seq_len = n_views = 1)."""
    if frames is None:
        return None
    if len(frames) != n or min(frames) < 1:
        raise ValueError("frames must hold one positive count per modality, got %r for %d modalities" % (frames, n))
    return (ctypes.c_int32 * n)(*[int(f) for f in frames])


def tokens_fwd(feats, pos, vel_w, vel_b, velocity, tok, drop_p=0.0, rng_state=None, rng_stream=0, frames=None):
    """feats[m] is [B * frames[m], S, S, C]; tok is [B, sum(frames) * 64, C]."""
    B = tok.shape[0]
    _, S, _, C = feats[0].shape
    for m, f in enumerate(feats):
        if f.shape[0] != B * (1 if frames is None else frames[m]):
            raise ValueError("modality %d holds %d frames for %d samples, expected %d per sample" % (m, f.shape[0], B, 1 if frames is None else frames[m]))
    if tok.shape[1] != 64 * (len(feats) if frames is None else sum(frames)):
        raise ValueError("token buffer has %d tokens, the frames need %d" % (tok.shape[1], 64 * (len(feats) if frames is None else sum(frames))))
    arr = _ptr_array(feats)
    if feats[0].dtype == BF16:   # bf16 features; the token matrix bf16 or (the transformers' residual stream) fp32
        _call("mmfn_tokens_fwd_bf16", arr, len(feats), _frames(frames, len(feats)), B, S, C, ptr(pos), ptr(vel_w), ptr(vel_b), ptr(velocity),
              ptr(tok), 0 if tok.dtype == BF16 else 1, float(drop_p), ptr(rng_state), rng_stream, stream())
        return tok
    _call("mmfn_tokens_fwd_f32", arr, len(feats), _frames(frames, len(feats)), B, S, C, ptr(pos), ptr(vel_w),
          ptr(vel_b), ptr(velocity), ptr(tok), float(drop_p), ptr(rng_state), rng_stream, stream())
    return tok


def tokens_bwd(gtok, velocity, dpos, dvel_w, dvel_b, drop_p=0.0, rng_state=None, rng_stream=0):
    B, T, C = gtok.shape
    need = lib().mmfn_tokens_bwd_workspace_bytes(T, C)
    _call("mmfn_tokens_bwd_bf16" if gtok.dtype == BF16 else "mmfn_tokens_bwd_f32", ptr(gtok), B, T, C, ptr(velocity), ptr(dpos), ptr(dvel_w), ptr(dvel_b), float(drop_p),
          ptr(rng_state), rng_stream, ptr(norm_workspace(gtok.device, need)), stream())


def _groups_ok(n_img, n_tok, T, m, frames):
    if frames < 1 or n_img != n_tok * frames or (m + frames) * 64 > T:
        raise ValueError("%d frames / %d samples / %d tokens do not fit token groups [%d, %d)" % (n_img, n_tok, T, m, m + frames))


def upsample_add_fwd(feat, tok, out, m, frames=1):
    """m: the modality's first token group, frames: its frames per sample (feat holds samples * frames maps)."""
    B, S, _, C = feat.shape
    T = tok.shape[1]
    _groups_ok(B, tok.shape[0], T, m, frames)
    _call("mmfn_upsample_add_fwd_bf16" if feat.dtype == BF16 else "mmfn_upsample_add_fwd_f32", ptr(feat), ptr(tok), ptr(out), B, S, C, T, m, frames, stream())
    return out


def upsample_adj(G, gtok, m, frames=1):
    B, S, _, C = G.shape
    T = gtok.shape[1]
    _groups_ok(B, gtok.shape[0], T, m, frames)
    _call("mmfn_upsample_adj_bf16" if G.dtype == BF16 else "mmfn_upsample_adj_f32", ptr(G), ptr(gtok), B, S, C, T, m, frames, stream())


def pool_bcast_add(G, gtok, dF, m, frames=1):
    B, S, _, C = G.shape
    T = gtok.shape[1]
    _groups_ok(B, gtok.shape[0], T, m, frames)
    if G.dtype == BF16:   # the token gradient bf16 or (the transformers' fp32 residual stream) fp32
        _call("mmfn_pool_bcast_add_bf16", ptr(G), ptr(gtok), 0 if gtok.dtype == BF16 else 1, ptr(dF), B, S, C, T, m, frames, stream())
        return dF
    _call("mmfn_pool_bcast_add_f32", ptr(G), ptr(gtok), ptr(dF), B, S, C, T, m, frames, stream())
    return dF


def gap_sum_fwd(feats, out, frames=None):
    """out[B, C] = sum over modalities and over each sample's frames of the spatial mean."""
    _, H, W, C = feats[0].shape
    B = out.shape[0]
    _call("mmfn_gap_sum_fwd_bf16" if feats[0].dtype == BF16 else "mmfn_gap_sum_fwd_f32", _ptr_array(feats), len(feats), _frames(frames, len(feats)), B, H * W, C,
          ptr(out), stream())
    return out


def gap_sum_bwd(g, outs, frames=None):
    _, H, W, C = outs[0].shape
    B = g.shape[0]
    _call("mmfn_gap_sum_bwd_bf16" if outs[0].dtype == BF16 else "mmfn_gap_sum_bwd_f32", ptr(g), _ptr_array(outs), len(outs), _frames(frames, len(outs)), B, H * W, C,
          stream())


def transpose(inp, out, B, R, Cc):
    name = {(torch.float32, torch.float32): "mmfn_transpose_f32", (torch.float32, BF16): "mmfn_transpose_f32_to_bf16",
            (BF16, torch.float32): "mmfn_transpose_bf16_to_f32"}[(inp.dtype, out.dtype)]
    _call(name, ptr(inp), ptr(out), B, R, Cc, stream())
    return out


# ---------------------------------------------------------------- attention
def attention_fwd(q, k, v, ld, o, ldo, lse, B, T, NH, HS, scale, kv_len=None, drop_p=0.0, rng_state=None, rng_stream=0):
    _call("mmfn_attention_fwd_bf16" if q.dtype == BF16 else "mmfn_attention_fwd_f32", ptr(q), ptr(k), ptr(v), ld, ptr(o), ldo, ptr(lse), B, T, NH, HS, float(scale),
          ptr(kv_len), float(drop_p), ptr(rng_state), rng_stream, stream())
    return o


def attention_bwd(q, k, v, ld, o, dO, ldo, lse, delta, dq, dk, dv, ldg, B, T, NH, HS, scale, kv_len=None, drop_p=0.0,
                  rng_state=None, rng_stream=0):
    _call("mmfn_attention_bwd_bf16" if q.dtype == BF16 else "mmfn_attention_bwd_f32", ptr(q), ptr(k), ptr(v), ld, ptr(o), ptr(dO), ldo, ptr(lse), ptr(delta), ptr(dq),
          ptr(dk), ptr(dv), ldg, B, T, NH, HS, float(scale), ptr(kv_len), float(drop_p), ptr(rng_state), rng_stream, stream())


# ---------------------------------------------------------------- fused GPT block (narrow fusion transformers)
GPT_ROWS = 32   # token rows per workgroup of the row-block kernels = rows per LayerNorm partial row
# bf16 mode: the fields of mmfn_gpt_block_desc that are bf16 (GEMM operands in HBM and the weight shadows); everything else stays fp32
GPT_BF16_FIELDS = frozenset(("wqkv", "wproj", "w1", "w2", "a", "qkv", "o", "a2", "h", "gd", "gh", "gd2", "go", "dqkv", "gd_below"))


def gpt_block_supported(C, NH, T):
    return lib().mmfn_gpt_block_supported(C, NH, T) == 0


def gpt_block_rows_supported(C, T):
    return lib().mmfn_gpt_block_rows_supported(C, T) == 0


def gpt_block_desc(B, T, C, NH, eps=1e-5, attn_pdrop=0.0, resid_pdrop=0.0, rng_state=None, rng_stream=0, rng_stream_below=0,
                   below_colsum=False, **tensors):
    """mmfn_gpt_block_desc (include/mmfn_hip.h): tensors by field name (contiguous); missing fields stay NULL.  fp32 everywhere on the
    fp32 path; in the bf16 mode the GEMM operands (GPT_BF16_FIELDS, and the weight shadows) are bf16.  The returned structure
    keeps its tensors alive and remembers whether it describes the bf16 mode (`d.bf16`)."""
    d = GptBlockDesc()
    bf16 = False
    for name, t in tensors.items():
        if name not in GptBlockDesc._PTRS:
            raise KeyError(name)
        if t is not None:
            assert t.is_contiguous(), name
            if t.dtype == BF16:
                assert name in GPT_BF16_FIELDS, name
                bf16 = True
            else:
                assert t.dtype == torch.float32, name
        setattr(d, name, ptr(t))
    if bf16:   # a descriptor is all-fp32 or has every operand it names in bf16
        assert all(t is None or t.dtype == BF16 for n, t in tensors.items() if n in GPT_BF16_FIELDS), "mixed operand precisions"
    d.bf16 = bf16
    d.rng_state = ptr(rng_state)
    d.B, d.T, d.C, d.NH = B, T, C, NH
    d.attn_pdrop, d.resid_pdrop, d.eps = float(attn_pdrop), float(resid_pdrop), float(eps)
    d.rng_stream, d.rng_stream_below, d.below_colsum = rng_stream, rng_stream_below, 1 if below_colsum else 0
    d._keep = (tensors, rng_state)
    return d


def gpt_block_attn_fwd(d):
    """ln1 -> key / query / value -> attention of one transformer block, one launch (model_vec.py:96-105,126).  fp32 path only."""
    assert not d.bf16
    _call("mmfn_gpt_block_attn_fwd_f32", ctypes.addressof(d), stream())


def gpt_block_mlp_fwd(d):
    """proj (+ residual) -> ln2 -> mlp.0 -> ReLU -> mlp.2 (+ residual) of one transformer block, one launch (model_vec.py:107-108,126-131)."""
    _call("mmfn_gpt_block_mlp_fwd_bf16" if d.bf16 else "mmfn_gpt_block_mlp_fwd_f32", ctypes.addressof(d), stream())


def gpt_block_bwd_rows(upper, lower):
    """The row-local backward between two attention backward passes: upper block's qkv dgrad + ln1 backward, lower block's mlp /
    ln2 / proj dgrads.  Either may be None."""
    bf16 = (upper if upper is not None else lower).bf16
    assert upper is None or lower is None or upper.bf16 == lower.bf16
    _call("mmfn_gpt_block_bwd_rows_bf16" if bf16 else "mmfn_gpt_block_bwd_rows_f32", None if upper is None else ctypes.addressof(upper),
          None if lower is None else ctypes.addressof(lower), stream())


def lane0_attention_fwd(qkv, kv_len, B, L, heads, head_dim, scale, att0, prob):
    """VectorNet lane attention for query 0 only (the only row the encoder consumes): any number of lanes."""
    _call("mmfn_lane0_attention_fwd_f32", ptr(qkv), ptr(kv_len), B, L, heads, head_dim, float(scale), ptr(att0), ptr(prob), stream())
    return att0


def lane0_attention_bwd(qkv, prob, g_att0, kv_len, B, L, heads, head_dim, scale, dqkv):
    _call("mmfn_lane0_attention_bwd_f32", ptr(qkv), ptr(prob), ptr(g_att0), ptr(kv_len), B, L, heads, head_dim, float(scale),
          ptr(dqkv), stream())
    return dqkv


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


def adamw_groups(p, g, m, v, step, hyper, n_groups, group_of=None, n=None):
    """AdamW with the hyper-parameter rows (and optional per-float4 group ids) in device memory."""
    n = p.numel() if n is None else n
    _call("mmfn_adamw_groups_f32", ptr(p), ptr(g), ptr(m), ptr(v), n, ptr(group_of), ptr(hyper), n_groups, ptr(step), stream())


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


# ---------------------------------------------------------------- bf16 weight shadows
def make_ln_fold_table(entries, device):
    """entries: [(W [N,K], gamma [K], beta [K], bias [N] or None, Wf [N,K], c1 [N], c2 [N])] -> (device table, n, total rows) for
    ln_fold_weights."""
    import struct
    blob, row0 = b"", 0
    for W, gamma, beta, bias, Wf, c1, c2 in entries:
        N, K = W.shape
        assert W.is_contiguous() and Wf.is_contiguous() and Wf.shape == W.shape and gamma.numel() == K and c1.numel() == N
        blob += struct.pack("<QQQQQQQiiq", W.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 0 if bias is None else bias.data_ptr(),
                            Wf.data_ptr(), c1.data_ptr(), c2.data_ptr(), N, K, row0)
        row0 += N
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    return t, len(entries), row0


def ln_fold_weights(table, n, total_rows):
    """The folded operands of every (LayerNorm -> Linear) pair of the step in one launch (once per forward: weights change at the
    optimizer step)."""
    _call("mmfn_ln_fold_weights_f32", ptr(table), n, total_rows, stream())


def cast_to_bf16(src, dst):
    _call("mmfn_cast_f32_to_bf16", ptr(src), ptr(dst), src.numel(), stream())
    return dst


def cast_to_f32(src, dst):
    _call("mmfn_cast_bf16_to_f32", ptr(src), ptr(dst), src.numel(), stream())
    return dst


def make_shadow_table(entries, device):
    """entries: [(src fp32 storage tensor [R, T, C] or [R, C], dst bf16 flat tensor)] -> (device table, n, total tiles) for
    shadow_transpose (dst[c][t][r] = src[r][t][c])."""
    import struct
    blob, tile0 = b"", 0
    for src, dst in entries:
        if src.dim() == 2:
            R, T, C = src.shape[0], 1, src.shape[1]
        else:
            R, T, C = src.shape[0], src.numel() // (src.shape[0] * src.shape[-1]), src.shape[-1]
        tiles_c = (C + 31) // 32
        blob += struct.pack("<QQiiiiq", src.data_ptr(), dst.data_ptr(), R, T, C, tiles_c, tile0)
        tile0 += T * ((R + 31) // 32) * tiles_c
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    return t, len(entries), tile0


def shadow_transpose(table, n, total):
    _call("mmfn_shadow_transpose_bf16", ptr(table), n, total, stream())


# ---------------------------------------------------------------- 7x7 stems: explicit im2col + plain GEMM
STEM_IM2COL = True   # the 7x7 stems as explicit im2col + plain GEMM (engine.ConvBN.stem_conv)


def im2col_small(x, col, kh, kw, stride, pad):
    B, H, W, Cin = x.shape
    _call("mmfn_im2col_small", ptr(x), ptr(col), 1 if col.dtype == BF16 else 0, B, H, W, Cin, kh, kw, stride, pad, col.shape[1], stream())
    return col


def repitch_rows(src, dst, R, K, ps, pd):
    _call("mmfn_repitch_rows", ptr(src), ptr(dst), 1 if dst.dtype == BF16 else 0, R, K, ps, pd, stream())
    return dst
