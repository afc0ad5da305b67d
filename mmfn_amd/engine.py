"""Static execution plan of the MMFN training step on one MI355X.

There is no autograd tape and no tracing compiler: forward and backward of every layer are
written out explicitly as sequences of HIP launches (mmfn_amd.ops -> libmmfn_hip.so) over
buffers that are allocated once per batch size.  Nothing allocates or synchronises during a
step, so a whole step can be captured into a hipGraph and replayed (mmfn_amd.model).

Data layout: feature maps are NHWC [B,H,W,C]; transformer tokens are rows of a [B*T, C] matrix,
which is the same memory as an 8x8 NHWC map, so the reference's permute/contiguous copies
around each GPT (model_vec.py:228,240-244) do not exist here.

Backward conventions: `bwd` methods receive dL/d(output) buffers and write parameter gradients
straight into the flat gradient buffer (each parameter has exactly one writer per step).
"""
import math

import os

import torch

from . import ops
from .ops import ACT_GELU, ACT_NONE, ACT_RELU


class Buffers(object):
    """Named, shape-keyed activation buffers: allocated on first use, reused on every later step."""

    def __init__(self, device):
        self.device = device
        self._bufs = {}

    def get(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("buffer %s%s first used during graph capture; run one eager step first" % (name, tuple(shape)))
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self._bufs.values())


class Ctx(object):
    """Per-call execution context."""

    def __init__(self, bufs, training, drop_p, rng_state, side=None, engine=None):
        self.bufs = bufs
        self.engine = engine
        self.wino_ready = False   # True: the filters of the registered Winograd layers were transformed by the grouped launch
        self.wino_names = frozenset()   # ... and the layers that launch covered
        self.training = training
        self.drop = drop_p if training else (0.0, 0.0, 0.0)  # (embd, attn, resid)
        self.rng_state = rng_state
        self.side = side  # side stream for work that is off the critical path (weight gradients), or None
        self.side2 = None  # a second one (lane 2's stream): the launch-bound transformers split their side work over both (GPT._offload_side)
        self.side2_used = False
        self.adt = engine.act_dtype if engine is not None else torch.float32   # dtype of activation / activation-gradient buffers
        self.bf16 = self.adt == torch.bfloat16
        self.folded = False  # eval forward over BatchNorm-folded filters (Engine.fold_batchnorm)

    def wino_u(self, name, w):
        """Engine-wide buffer for a layer's transformed filter U [36][Co][Ci]; registers the layer for the grouped transform."""
        eng = self.engine
        u = eng.wino_layers.get(name)
        if u is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("Winograd filter buffer %s first used during graph capture; run one eager step first" % name)
            u = (w, torch.empty(36 * w.shape[0] * w.shape[3], dtype=torch.float32, device=w.device))
            eng.wino_layers[name] = u
            eng.wino_table = None   # rebuilt before the next forward; THIS forward transforms the new layer's filter itself
        return u[1]

    def wino_in_table(self, name):
        """True when the grouped launch at the head of this forward transformed this layer's filter (the layer was registered
        when the table was built - a layer that registers later, e.g. for a new image size, is not)."""
        return self.wino_ready and name in self.wino_names

    def offload(self, fn):
        """Run fn (launches that only produce parameter gradients) on the side stream, ordered after everything
        enqueued so far on the current stream.  Inline when no side stream is configured."""
        if self.side is None:
            return fn()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side), ops.lane(1):
            fn()

    def fork_point(self):
        """An event at the current point of the current stream, for offload_at()."""
        if self.side is None:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def offload_at(self, ev, fn, second=False):
        """Like offload(), but ordered after the earlier fork_point() `ev` instead of after everything enqueued so far.
        Used to enqueue side work AFTER the chain's next kernel: when a captured graph is replayed, the first-captured child
        of a node stays on the node's hardware queue and later children move to other queues; capturing the chain's
        continuation first keeps the dependent chain on one queue (a queue hop costs 10-16 us of idle time)."""
        if self.side is None or ev is None:
            return fn()
        st, ln = (self.side2, 2) if (second and self.side2 is not None) else (self.side, 1)
        st.wait_event(ev)
        with torch.cuda.stream(st), ops.lane(ln):
            fn()
        if st is self.side2:
            self.side2_used = True

    def rejoin(self):
        """Current stream waits for all offloaded work (before its input buffers are reused)."""
        if self.side is None:
            return
        ev = torch.cuda.Event()
        ev.record(self.side)
        torch.cuda.current_stream().wait_event(ev)
        if self.side2_used:
            ev2 = torch.cuda.Event()
            ev2.record(self.side2)
            torch.cuda.current_stream().wait_event(ev2)
            self.side2_used = False


# ----------------------------------------------------------------------------- conv + BN
FUSE_BN_BWD_REDUCE = True   # see ConvBN.bwd16 (bf16 mode)
# fp32: a BatchNorm apply (+ residual + ReLU) whose consumer is an F(4x4) Winograd convolution runs inside that convolution's input
# transform (PendingBN).  A/B switch; 0 = every BatchNorm apply is its own launch (the round-3 step).
LAZY_BN_APPLY = os.environ.get("MMFN_LAZY_BN", "1") == "1"


# bf16 mode: the same laziness - a 3x3 stride-1 convolution whose input is the PendingBN of its producer runs as
# mmfn_conv3x3_halo_bf16 (csrc/conv16_halo.hip), which applies the producer's BatchNorm (+ skip + ReLU) while it stages its halo
# patch and writes the activation for the pixels it owns.  Measured per shape (tools/halo_bench.py, B = 32): 64 / 128 contraction
# channels 1.3-1.4 x the apply + implicit-GEMM pair; 256 / 512 channels are bound by the filter stream per 64-pixel tile and stay on
# the implicit GEMM (MMFN_HALO_MAX_K).  MMFN_HALO_CONV=0 restores round 5's launch sequence.
HALO_MAX_K = int(os.environ.get("MMFN_HALO_MAX_K", "128"))
# backward: 0 = implicit-GEMM data gradients (round 5), 1 = the halo kernel as the data gradient, 2 = also the BatchNorm backward's
# elementwise pass in its loader (ConvBN.bwd16).  Same-box A/B (vec B = 32, medians of 3 x 100 steps): 16.35 / 16.09 / 16.19 ms - the
# fused loader reads three tensors per patch element and makes the weight gradient wait for the data gradient: 1 is the default.
HALO_BWD = int(os.environ.get("MMFN_HALO_BWD", "1"))


class PendingBN(object):
    """The output of a ConvBN whose BatchNorm apply (+ residual) (+ ReLU) has not run: the convolution output and the batch
    statistics are in HBM, the activation y = [relu](bn(co) [+ res]) is not.  Two ways out:
      * consume(...) by an F(4x4) Winograd convolution - the apply runs inside its input transform (ops.conv2d_winograd(x_bn=...),
        csrc/winograd.hip wino4_input_bn_kernel); with want_y the transform also writes y (a block output, which the next
        block's skip connection and the backward's ReLU mask read), otherwise y never exists and the producer's backward
        recomputes the ReLU sign from co;
      * tensor(ctx): the plain apply launch (any other consumer: pooling, token kernels, a strided convolution)."""

    def __init__(self, owner, co, res, relu):
        self.owner, self.co, self.res, self.relu = owner, co, res, relu
        self.shape, self.dtype = co.shape, co.dtype
        self.y = None

    def _out(self, ctx):
        return ctx.bufs.get(self.owner.name + ".out", self.shape, ctx.adt if ctx.bf16 else torch.float32)

    def tensor(self, ctx):
        if self.y is None:
            o = self.owner
            M = self.co.numel() // o.cout
            self.y = self._out(ctx)
            ops.bn_apply(self.co.view(M, o.cout), self.y.view(M, o.cout), o.saved[3], o.saved[4], o.bn_w, o.bn_b, self.relu,
                         res=None if self.res is None else self.res.view(M, o.cout))
            o.saved[2] = self.y
        return self.y

    def consume(self, ctx, want_y):
        """-> the x_bn tuple of ops.conv2d_winograd; afterwards self.y is the written activation (want_y) or stays None."""
        o = self.owner
        if want_y and self.y is None:
            self.y = self._out(ctx)
            o.saved[2] = self.y
            y_out = self.y
        else:
            y_out = None
        return (self.res, o.saved[3], o.saved[4], o.bn_w, o.bn_b, self.relu, y_out)


def as_tensor(ctx, x):
    return x.tensor(ctx) if isinstance(x, PendingBN) else x


class ConvBN(object):
    """conv (bias-free) -> BatchNorm2d -> [+ residual] -> [ReLU]."""

    def __init__(self, name, layout, conv_name, bn_name, bn_mod, stride, pad):
        self.name = name
        self.w = layout.w(conv_name + ".weight")
        self.gw = layout.g(conv_name + ".weight")
        self.bn_w, self.bn_b = layout.w(bn_name + ".weight"), layout.w(bn_name + ".bias")
        self.g_bn_w, self.g_bn_b = layout.g(bn_name + ".weight"), layout.g(bn_name + ".bias")
        self.bn = bn_mod
        self.stride, self.pad = stride, pad
        self.cout = self.w.shape[0]
        self.conv_name = conv_name
        self.w16 = self.w16t = None   # bf16 shadows [Cout,KH,KW,Cin] / [Cin,KH,KW,Cout] (Engine._build_shadows, bf16 mode)

    # ---- the 7x7 stems (3 / 2 input channels) as explicit im2col + plain GEMM (csrc/misc.hip: mmfn_im2col_small)
    def is_stem(self):
        return self.w.shape[3] <= 4 and ops.STEM_IM2COL

    def stem_conv(self, ctx, x, co, stats=None, w=None, bias=None, relu=False):
        """co [B,OH,OW,Cout] = conv(x) through the patch matrix, which is kept for the weight gradient (stem_wgrad).
        w / bias / relu: the BatchNorm-folded filter and shift of an eval forward (Engine.fold_batchnorm)."""
        Co, KH, KW, Ci = self.w.shape
        K = KH * KW * Ci
        M = co.numel() // Co
        tile = 64 if co.dtype == torch.bfloat16 else 16
        KP = (K + tile - 1) // tile * tile
        col = ctx.bufs.get(self.name + ".col", (M, KP), co.dtype)
        wp = ctx.bufs.get(self.name + ".wpad", (Co, KP), co.dtype)
        ops.im2col_small(x, col, KH, KW, self.stride, self.pad)
        ops.repitch_rows(self.w if w is None else w, wp, Co, K, K, KP)
        if stats is not None:
            ops.linear_fwd(col, wp, None, out=co.view(M, Co), stats=stats)
        elif bias is not None:
            ops.linear_fwd(col, wp, bias, out=co.view(M, Co), relu=relu)
        else:
            ops.linear_fwd(col, wp, None, out=co.view(M, Co))
        self.saved_col = col
        return co

    def stem_wgrad(self, ctx, dco):
        Co, KH, KW, Ci = self.w.shape
        K = KH * KW * Ci
        col = self.saved_col
        KP = col.shape[1]
        dwp = ctx.bufs.get(self.name + ".dwpad", (Co, KP))
        ops.linear_dw(dco.view(-1, Co), col, out=dwp)
        ops.repitch_rows(dwp, self.gw, Co, K, KP, K)

    def fwd16(self, ctx, x, relu=True, res=None, lazy=False):
        """bf16 mode.  A trunk convolution (bf16 input): direct implicit GEMM on the bf16 MFMA pipe whose epilogue also emits
        the BatchNorm batch-statistics partial sums of its fp32 accumulators (no statistics pass over the output); a stem (fp32
        input from the ingest kernels, 3 / 2 channels): the fp32 convolution, fp32 output, and the BatchNorm apply is where the
        activation becomes bf16.  x may be the PendingBN of the producing ConvBN: a 3x3 stride-1 convolution of up to HALO_MAX_K
        channels then runs as mmfn_conv3x3_halo_bf16 and applies that BatchNorm (+ skip + ReLU) in its loader; lazy: return this
        layer's own PendingBN instead of launching the apply."""
        from . import ops16
        # the LDS-resident-patch kernel (3x3 stride 1): takes the producer's PendingBN as it is and applies it in its loader
        halo = x.dtype == torch.bfloat16 and x.shape[-1] <= HALO_MAX_K and ops16.halo_ok(tuple(x.shape), tuple(self.w.shape), self.stride, self.pad) > 0
        x_bn = None
        if isinstance(x, PendingBN):
            if halo and x.y is None and LAZY_BN_APPLY and not ctx.folded:
                pend = x
                x_bn = pend.consume(ctx, True)    # (res, mean, rstd, weight, bias, relu, y_out): y is written by this convolution
                xin, x = pend.co, pend.y
            else:
                x = xin = x.tensor(ctx)
        else:
            xin = x
        _, oshape = ops.conv_geom(x.shape, self.w.shape, self.stride, self.pad)
        M = oshape[0] * oshape[1] * oshape[2]
        stem = x.dtype == torch.float32
        if not ctx.training and ctx.folded and (not stem or self.is_stem()):
            # eval over BatchNorm-folded filters (Engine.fold_batchnorm; the closed-loop session): convolution + shift + skip + ReLU
            # in ONE launch of the implicit GEMM, the folded filter as a bf16 shadow
            assert x_bn is None
            wf, bf, wf16 = ctx.engine.folded[self.name]
            y = ctx.bufs.get(self.name + ".out", oshape, ctx.adt)
            if stem:
                self.stem_conv(ctx, x, y, w=wf, bias=bf, relu=relu)
            elif res is None:
                ops16.conv2d_fwd(x, wf16, self.stride, self.pad, y, bias=bf, relu=relu)
            else:
                ops16.conv2d_fwd(x, wf16, self.stride, self.pad, y, bias=bf, res=res.view(M, self.cout), ldr=self.cout, relu_last=relu)
            return y
        col_stem = stem and self.is_stem()   # patch-matrix form: the stem joins the bf16 pipeline (bf16 conv output, statistics from the epilogue)
        co = ctx.bufs.get(self.name + ".conv", oshape, torch.float32 if (stem and not col_stem) else ctx.adt)
        mean = ctx.bufs.get(self.name + ".mean", (self.cout,))
        rstd = ctx.bufs.get(self.name + ".rstd", (self.cout,))
        bn = self.bn
        if col_stem:
            if ctx.training:
                ws = ops.norm_workspace(x.device, 2 * ((M + 63) // 64) * 2 * self.cout * 8)
                self.stem_conv(ctx, x, co, stats=ws)
                rows = ops16.gemm_stats_rows(ops16.G16_NT, M, self.cout, self.saved_col.shape[1])
                ops.bn_finalize_stats(ws, rows, M, self.cout, mean, rstd, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                      bn.eps, bn.momentum)
            else:
                self.stem_conv(ctx, x, co)
        elif stem:
            if ctx.training:
                ops.conv2d_fwd_bn_stats(x, self.w, self.stride, self.pad, co, mean, rstd, bn.running_mean, bn.running_var,
                                        bn.num_batches_tracked, bn.eps, bn.momentum)
            else:
                ops.conv2d_fwd(x, self.w, self.stride, self.pad, out=co)
        elif halo:
            ws = ops.norm_workspace(x.device, 2 * ((M + 63) // 64) * 2 * self.cout * 8) if ctx.training else None
            bn_apply = None if x_bn is None else (x_bn[1], x_bn[2], x_bn[3], x_bn[4], x_bn[0], x_bn[5], x_bn[6])
            rows = ops16.conv3x3_halo(xin, self.w16, co, stats=ws, bn_apply=bn_apply)
            if ctx.training:
                ops.bn_finalize_stats(ws, rows, M, self.cout, mean, rstd, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                      bn.eps, bn.momentum)
        elif ctx.training:
            ws = ops.norm_workspace(x.device, 2 * ((M + 63) // 64) * 2 * self.cout * 8)   # sized for the smallest tile
            ops16.conv2d_fwd(x, self.w16, self.stride, self.pad, co, stats=ws)
            rows = ops16.conv_stats_rows(x.shape, self.w.shape, self.stride, self.pad)   # AFTER the launch: autotune may add the entry
            ops.bn_finalize_stats(ws, rows, M, self.cout, mean, rstd, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                  bn.eps, bn.momentum)
        else:
            ops16.conv2d_fwd(x, self.w16, self.stride, self.pad, co)
        if not ctx.training:
            ops.bn_eval_prepare(bn.running_mean, bn.running_var, mean, rstd, bn.eps)
        self.saved = [x, co, None, mean, rstd, relu]   # saved[2]: the activation, written by PendingBN.tensor() or by the consumer's loader
        pend = PendingBN(self, co, res, relu)
        if lazy and LAZY_BN_APPLY and HALO_MAX_K > 0:
            return pend
        return pend.tensor(ctx)

    def bwd16(self, ctx, g, need_dx=True, ge_out=None, dx_res=None, mask_y=None, emit=None):
        """emit: the ConvBN whose BatchNorm receives THIS call's data gradient as its output gradient (the layer below on the
        dependent chain): the data-gradient GEMM's epilogue then also forms the two reductions of that BatchNorm's backward
        (sum ge, sum ge * xhat) from the tile it is storing, so the consumer skips its reduction pass over g, y and x."""
        from . import ops16
        x, co, y, mean, rstd, relu = self.saved
        M = co.numel() // self.cout
        dco = ctx.bufs.get(self.name + ".dconv", co.shape, co.dtype)
        ymask = (y if mask_y is None else mask_y).view(M, self.cout) if relu else None
        pre, self._pre = getattr(self, "_pre", None), None
        # data gradient through the LDS-resident-patch kernel (3x3 stride 1, contraction = this layer's output channels); with the
        # reductions already emitted by the layer above (`pre`) the BatchNorm backward's elementwise pass runs in ITS loader: the
        # chain is finalize -> data gradient (which writes dco for the weight gradient) instead of finalize -> apply -> data gradient
        halo_dg = need_dx and HALO_BWD >= 1 and x.dtype == torch.bfloat16 and co.dtype == torch.bfloat16 and self.cout <= HALO_MAX_K and \
            ops16.halo_ok(tuple(co.shape), tuple(self.w16t.shape), self.stride, self.pad) > 0
        fuse = halo_dg and HALO_BWD >= 2 and pre is not None and mask_y is None
        bn_bwd_arg = None
        if fuse:
            means = ctx.bufs.get(self.name + ".bnmeans", (2, self.cout))
            ops16.bn_bwd_finalize(pre[0], pre[1], M, self.cout, self.g_bn_w, self.g_bn_b, means)
            bn_bwd_arg = (mean, rstd, self.bn_w, means, y if relu else None, co, dco, ge_out)
        elif pre is not None and mask_y is None and co.dtype == torch.bfloat16:
            ops16.bn_bwd_partials(pre[0], pre[1], g.view(M, self.cout), ymask, co.view(M, self.cout), mean, rstd, self.bn_w,
                                  dco.view(M, self.cout), self.g_bn_w, self.g_bn_b,
                                  ge_out=None if ge_out is None else ge_out.view(M, self.cout))
        else:
            ops.bn_bwd(g.view(M, self.cout), ymask, co.view(M, self.cout), mean, rstd, self.bn_w, dco.view(M, self.cout),
                       self.g_bn_w, self.g_bn_b, ge_out=None if ge_out is None else ge_out.view(M, self.cout))
        if x.dtype == torch.float32:   # stem: no data gradient
            if self.is_stem():
                self.stem_wgrad(ctx, dco)
            else:
                ops.conv2d_wgrad(dco, x, tuple(self.w.shape), self.stride, self.pad, out=self.gw)
            return None
        if not fuse:
            ops16.conv2d_wgrad(dco, x, tuple(self.w.shape), self.stride, self.pad, self.gw)
        if not need_dx:
            return None
        dx = ctx.bufs.get(self.name + ".dx", x.shape, ctx.adt)
        cin = x.shape[-1]
        extra = {}
        if emit is not None and FUSE_BN_BWD_REDUCE and emit.saved[1].dtype == torch.bfloat16:
            ex, eco, ey, emean, erstd, erelu = emit.saved
            Mx = x.shape[0] * x.shape[1] * x.shape[2]
            part = ctx.bufs.get(emit.name + ".bnpart", (ops16.max_stats_rows(Mx), 2, cin), torch.float64)
            extra = dict(stats=part, stats_mode=2, bn=(ey if erelu else None, eco, emean, erstd))
        if halo_dg:
            rows = ops16.conv3x3_halo(g if fuse else dco, self.w16t, dx, flip=True, out_res=dx_res, stats=extra.get("stats"), stats_mode=2,
                                      bn2=extra.get("bn"), bn_bwd=bn_bwd_arg)
            if "stats" in extra:
                emit._pre = (extra["stats"], rows)
            if fuse:   # dco exists only now: the loader of the data gradient wrote it
                ops16.conv2d_wgrad(dco, x, tuple(self.w.shape), self.stride, self.pad, self.gw)
            return dx
        if dx_res is not None:
            extra.update(res=dx_res.view(-1, cin), ldr=cin)
        ops16.conv2d_dgrad(dco, self.w16t, tuple(x.shape), tuple(self.w.shape), self.stride, self.pad, dx, **extra)
        if "stats" in extra:   # (after the launch: the autotuner may just have added the shape's table entry)
            g_, _ = ops.conv_geom(tuple(x.shape), tuple(self.w.shape), self.stride, self.pad)
            emit._pre = (extra["stats"], ops16.gemm_stats_rows(ops16.G16_CONV_DGRAD, Mx, cin, g_[6] * g_[7] * g_[5], g_))
        return dx

    def fwd(self, ctx, x, relu=True, res=None, lazy=False, want_x=True):
        """x: an NHWC tensor or the PendingBN of the producing ConvBN.  lazy: return a PendingBN instead of applying the BatchNorm
        (the caller hands it to the next convolution, or calls .tensor()).  want_x (x pending): the producer's activation is
        needed as a tensor as well (it is a block output)."""
        if ctx.bf16:
            return self.fwd16(ctx, x, relu, res, lazy)
        x_bn = None
        if isinstance(x, PendingBN):
            # the producer's BatchNorm apply inside this convolution's input transform: F(4x4) forward, and - when training - a
            # backward that never reads x (transformed input kept, adjoint data gradient)
            wshape, xs = self.w.shape, x.shape
            ok = LAZY_BN_APPLY and not ctx.folded and ops.winograd_f4_ok(xs, wshape, self.stride, self.pad) and x.y is None
            if ok and ctx.training:
                ok = ops.winograd_wgrad_ok(xs, wshape, self.stride, self.pad) and ops.winograd_adjoint_ok(xs, wshape, self.stride, self.pad)
            if ok:
                pend = x
                x_bn = pend.consume(ctx, want_x)
                x = pend.co      # what the transform reads; same shape as the activation
            else:
                x = x.tensor(ctx)
        B = x.shape[0]
        _, oshape = ops.conv_geom(x.shape, self.w.shape, self.stride, self.pad)
        co = ctx.bufs.get(self.name + ".conv", oshape)
        keep_v = keep_u = None
        if ctx.training and ops.winograd_ok(x.shape, self.w.shape, self.stride, self.pad, {}) and \
                ops.winograd_wgrad_ok(x.shape, self.w.shape, self.stride, self.pad):
            # the transformed input of the Winograd path is what the weight gradient needs again: keep it per layer
            keep_v = ctx.bufs.get(self.name + ".winoV", (ops.winograd_v_numel(x.shape),))
            if ops.winograd_adjoint_ok(x.shape, self.w.shape, self.stride, self.pad):
                # ... and the transformed filter is what the data gradient (adjoint of this forward) needs again.  The
                # buffer is engine-wide (it does not depend on the batch): Engine.forward transforms all of them in one launch
                keep_u = ctx.wino_u(self.name, self.w)
        M = oshape[0] * oshape[1] * oshape[2]
        if not ctx.training and ctx.folded:
            assert x_bn is None
            # eval with BatchNorm folded into the filter (Engine.fold_batchnorm): convolution + shift + skip + ReLU in ONE launch,
            # as a direct implicit GEMM (at batch 1 the Winograd form - transform, 36-batch GEMM, transform - measured slower: 4.99 vs
            # 4.59 ms per tick)
            wf, bf = ctx.engine.folded[self.name][:2]
            y = ctx.bufs.get(self.name + ".out", oshape)
            if res is None:
                ops.conv2d_fwd(x, wf, self.stride, self.pad, out=y, bias=bf, relu=relu)
            else:
                ops.conv2d_fwd(x, wf, self.stride, self.pad, out=y, bias=bf, res=res.view(M, self.cout), ldr=self.cout, relu_last=relu)
            return y
        co2 = co.view(M, self.cout)
        mean = ctx.bufs.get(self.name + ".mean", (self.cout,))
        rstd = ctx.bufs.get(self.name + ".rstd", (self.cout,))
        if self.is_stem():
            self.stem_conv(ctx, x, co)
            if ctx.training:
                ops.bn_train_stats(co2, mean, rstd, self.bn.running_mean, self.bn.running_var, self.bn.num_batches_tracked, self.bn.eps,
                                   self.bn.momentum)
            else:
                ops.bn_eval_prepare(self.bn.running_mean, self.bn.running_var, mean, rstd, self.bn.eps)
        elif ctx.training:
            ops.conv2d_fwd_bn_stats(x, self.w, self.stride, self.pad, co, mean, rstd, self.bn.running_mean, self.bn.running_var,
                                    self.bn.num_batches_tracked, self.bn.eps, self.bn.momentum, keep_v=keep_v, keep_u=keep_u,
                                    u_ready=keep_u is not None and ctx.wino_in_table(self.name), x_bn=x_bn)
        else:
            ops.conv2d_fwd(x, self.w, self.stride, self.pad, out=co, x_bn=x_bn)
            ops.bn_eval_prepare(self.bn.running_mean, self.bn.running_var, mean, rstd, self.bn.eps)
        # saved[0]: the input - with x_bn the producer's convolution output stands in: only its SHAPE is read by the backward
        # (the Winograd weight gradient uses the kept transformed input); saved[2]: the activation, None until / unless written
        self.saved = [x, co, None, mean, rstd, relu]
        self.saved_v = keep_v
        self.saved_u = keep_u
        self.x_is_standin = x_bn is not None
        pend = PendingBN(self, co, res, relu)
        if lazy and LAZY_BN_APPLY:
            return pend
        return pend.tensor(ctx)

    def bwd(self, ctx, g, need_dx=True, ge_out=None, dx_res=None, mask_y=None, emit=None):
        """g: dL/dy.  ge_out receives the ReLU-masked g (the residual branch's gradient).
        dx_res is added to dx (gradient arriving at x from another branch).
        mask_y overrides the tensor whose sign gates the ReLU (when y was further modified).
        emit (bf16 mode): see bwd16."""
        if ctx.bf16:
            return self.bwd16(ctx, g, need_dx, ge_out, dx_res, mask_y, emit)
        x, co, y, mean, rstd, relu = self.saved
        M = co.numel() // self.cout
        dco = ctx.bufs.get(self.name + ".dconv", co.shape)
        ymask = relu_bias = None
        if relu:
            ym = y if mask_y is None else mask_y
            if ym is not None:
                ymask = ym.view(M, self.cout)
            else:
                # the forward applied this BatchNorm + ReLU inside the next convolution's input transform and never wrote its
                # output (PendingBN): the backward kernels recompute the sign of bn(co) from co
                relu_bias = self.bn_b
        u = getattr(self, "saved_u", None)
        if need_dx and u is not None:
            # weight and data gradient together in the Winograd domain (shared A dy A^T), and dy = the BatchNorm backward of g
            # is formed inside that transform: only the two reductions run as kernels of their own, dy never goes to HBM
            means = ctx.bufs.get(self.name + ".bnmeans", (2, self.cout))
            ops.bn_bwd_reduce(g.view(M, self.cout), ymask, co.view(M, self.cout), mean, rstd, self.g_bn_w, self.g_bn_b, means,
                              relu_wb=None if relu_bias is None else (self.bn_w, relu_bias))
            dx = ctx.bufs.get(self.name + ".dx", x.shape)
            ops.conv2d_bwd_winograd(dco, x, u, self.gw, dx, v=getattr(self, "saved_v", None), res=dx_res,
                                    bn=(g, None if ymask is None else ymask.view(g.shape), co, mean, rstd, self.bn_w, relu_bias, means,
                                        ge_out))
            return dx
        assert not getattr(self, "x_is_standin", False), "%s: the direct backward needs the input activation" % self.name
        ops.bn_bwd(g.view(M, self.cout), ymask, co.view(M, self.cout), mean, rstd, self.bn_w, dco.view(M, self.cout),
                   self.g_bn_w, self.g_bn_b, ge_out=None if ge_out is None else ge_out.view(M, self.cout), relu_bias=relu_bias)
        if self.is_stem() and not need_dx:
            self.stem_wgrad(ctx, dco)
            return None
        ops.conv2d_wgrad(dco, x, tuple(self.w.shape), self.stride, self.pad, out=self.gw, v=getattr(self, "saved_v", None))
        if not need_dx:
            return None
        dx = ctx.bufs.get(self.name + ".dx", x.shape)
        if dx_res is None:
            ops.conv2d_dgrad(dco, self.w, tuple(x.shape), self.stride, self.pad, out=dx)
        else:
            cin = x.shape[-1]
            ops.conv2d_dgrad(dco, self.w, tuple(x.shape), self.stride, self.pad, out=dx, res=dx_res, ldr=cin)
        return dx


class BasicBlock(object):
    def __init__(self, name, layout, prefix, mod):
        self.name = name
        s = mod.stride
        self.c1 = ConvBN(name + ".c1", layout, prefix + ".conv1", prefix + ".bn1", mod.bn1, s, 1)
        self.c2 = ConvBN(name + ".c2", layout, prefix + ".conv2", prefix + ".bn2", mod.bn2, 1, 1)
        self.down = None
        if mod.downsample is not None:
            self.down = ConvBN(name + ".down", layout, prefix + ".downsample.0", prefix + ".downsample.1",
                               mod.downsample[1], s, 0)

    def fwd(self, ctx, x):
        """x: tensor, or the PendingBN of the previous block's output.  Returns the PendingBN of this block's output (fp32 path;
        the bf16 mode applies every BatchNorm eagerly): the first BatchNorm + ReLU is applied inside the second convolution's
        input transform and its output is never written; the second BatchNorm + skip + ReLU inside the NEXT block's first
        convolution, which also writes the block output (its own skip connection reads it)."""
        y1 = self.c1.fwd(ctx, x, relu=True, lazy=True, want_x=True)
        x = as_tensor(ctx, x)     # written by c1's input transform if it was pending, else by the apply launch
        skip = x if self.down is None else self.down.fwd(ctx, x, relu=False)
        return self.c2.fwd(ctx, y1, relu=True, res=skip, lazy=True, want_x=False)

    def bwd(self, ctx, g, mask_y=None, emit=None):
        """emit: the ConvBN (the previous block's second convolution) whose BatchNorm this block's input gradient enters."""
        ge = ctx.bufs.get(self.name + ".ge", g.shape, g.dtype)
        g_y1 = self.c2.bwd(ctx, g, ge_out=ge, mask_y=mask_y, emit=self.c1)
        if self.down is None:
            return self.c1.bwd(ctx, g_y1, dx_res=ge, emit=emit)
        dx_skip = self.down.bwd(ctx, ge)
        return self.c1.bwd(ctx, g_y1, dx_res=dx_skip, emit=emit)


class ResNetTrunk(object):
    def __init__(self, name, layout, prefix, mod, with_stem=True, first_layer=1):
        self.name = name
        self.stem = ConvBN(name + ".stem", layout, prefix + ".conv1", prefix + ".bn1", mod.bn1, 2, 3) if with_stem else None
        self.layers = {}
        for li in range(first_layer, 5):
            seq = getattr(mod, "layer%d" % li)
            self.layers[li] = [BasicBlock("%s.l%d.%d" % (name, li, j), layout, "%s.layer%d.%d" % (prefix, li, j), blk)
                               for j, blk in enumerate(seq)]

    def convbns(self):
        if self.stem is not None:
            yield self.stem
        for li in sorted(self.layers):
            for blk in self.layers[li]:
                yield blk.c1
                yield blk.c2
                if blk.down is not None:
                    yield blk.down

    def stem_fwd(self, ctx, x):
        y = self.stem.fwd(ctx, x, relu=True)
        B, H, W, C = y.shape
        p = ctx.bufs.get(self.name + ".pool", (B, (H + 1) // 2, (W + 1) // 2, C), y.dtype)
        idx = ctx.bufs.get(self.name + ".poolidx", p.shape, torch.uint8)
        ops.maxpool_fwd(y, p, idx)
        self.pool_saved = (y, idx)
        return p

    def stem_bwd(self, ctx, g):
        y, idx = self.pool_saved
        gy = ctx.bufs.get(self.name + ".dpool", y.shape, y.dtype)
        ops.maxpool_bwd(g, idx, gy)
        self.stem.bwd(ctx, gy, need_dx=False)

    def layer_fwd(self, ctx, li, x):
        for blk in self.layers[li]:
            x = blk.fwd(ctx, x)
        return as_tensor(ctx, x)

    def layer_bwd(self, ctx, li, g, mask_y=None):
        blocks = self.layers[li]
        for i in range(len(blocks) - 1, -1, -1):
            g = blocks[i].bwd(ctx, g, mask_y=mask_y if i == len(blocks) - 1 else None, emit=blocks[i - 1].c2 if i > 0 else None)
        return g


# ----------------------------------------------------------------------------- Linear helper
class Linear(object):
    def __init__(self, layout, prefix, bias=True):
        self.w, self.gw = layout.w(prefix + ".weight"), layout.g(prefix + ".weight")
        self.b = layout.w(prefix + ".bias") if bias else None
        self.gb = layout.g(prefix + ".bias") if bias else None
        self.name = prefix + ".weight"
        self.w16 = self.w16t = None   # bf16 shadows [out, in] / [in, out] (Engine._build_shadows, bf16 mode)


DEFER_LN_REDUCTIONS = True   # see LayerNorm.bwd


class LayerNorm(object):
    def __init__(self, name, layout, prefix):
        self.name = name
        self.w, self.b = layout.w(prefix + ".weight"), layout.w(prefix + ".bias")
        self.gw, self.gb = layout.g(prefix + ".weight"), layout.g(prefix + ".bias")

    def fwd(self, ctx, x, act=ACT_NONE, out=None):
        M, C = x.shape
        y = ctx.bufs.get(self.name + ".y", (M, C), x.dtype) if out is None else out
        mean, rstd = ctx.bufs.get(self.name + ".mu", (M,)), ctx.bufs.get(self.name + ".rs", (M,))
        ops.layernorm_fwd(x, self.w, self.b, y, mean, rstd, act)
        self.saved = (x, mean, rstd, act)
        return y

    def bwd(self, ctx, g, dres=None, out=None, dropped=None, drop_p=0.0, rng_stream=0, colsum=None, defer=False, collect=None):
        """defer: the caller rejoins the side stream before the gradients are consumed (GPT.bwd).  dropped: buffer that receives dx with the dropout mask (p, stream) of the branch consuming dx applied.
        colsum: [C] gradient buffer that receives the column sums of that tensor (the consuming Linear's bias gradient)."""
        x, mean, rstd, act = self.saved
        dx = ctx.bufs.get(self.name + ".dx", x.shape, x.dtype) if out is None else out
        rng = ctx.rng_state if dropped is not None else None
        if (defer is True or isinstance(defer, list)) and ctx.side is not None and DEFER_LN_REDUCTIONS:
            # the chain only needs dx: the reduction of the per-block partial rows into the weight / bias gradients (and the
            # consuming Linear's bias gradient) goes to the side stream, like the weight-gradient GEMMs (0.8 ms of ~11 us
            # launches per step off the transformers' dependent chain); own partial buffer, it must outlive this call
            M, C = x.shape
            rows = ops.layernorm_bwd_rows(M)
            part = ctx.bufs.get(self.name + ".part", (rows, 3 if colsum is not None else 2, C))
            ops.layernorm_bwd_partial(g, x, self.w, self.b, mean, rstd, dx, part, act, dres=dres, dx_dropped=dropped, drop_p=drop_p,
                                      rng_state=rng, rng_stream=rng_stream, want_colsum=colsum is not None)
            fin = lambda: ops.layernorm_bwd_finalize(part, rows, C, self.gw, self.gb, colsum)
            if collect is not None:
                collect.append((part, self.gw, self.gb, colsum, rows, C))   # GPT.bwd: one batched launch for all of them at its end
            elif isinstance(defer, list):
                defer.append(fin)      # the caller decides where the reductions run (GPT.bwd: one fork per block / none)
            else:
                ctx.offload(fin)
            return dx
        ops.layernorm_bwd(g, x, self.w, self.b, mean, rstd, dx, self.gw, self.gb, act, dres=dres, dx_dropped=dropped,
                          drop_p=drop_p, rng_state=rng, rng_stream=rng_stream, dx_colsum=colsum)
        return dx


# ----------------------------------------------------------------------------- GPT fusion transformer
# fp32 path: ln1 -> key/query/value and ln2 -> mlp.0 as ONE launch each (MMFN_EPI_LN_FOLD: the LayerNorm folded into the GEMM,
# include/mmfn_hip.h mmfn_gemm_desc.ln_c1); in a training step the normalised tensor that only the weight gradient needs is
# recomputed on the side stream in the backward.  MMFN_LN_FOLD = "eval" (default): eval-mode forwards only - validation and the
# batch-1 closed-loop tick, where 64 launches less on a dependent chain of ~330 are worth 4 % (4.05 vs 4.22 ms per tick);
# "1": training too - measured SLOWER there (32.01 vs 31.74 ms per step, DESIGN.md section 5: the folded GEMMs cost what the
# LayerNorm launches saved, and the recomputation competes with the backward's side work); "0": never.
LN_FOLD = os.environ.get("MMFN_LN_FOLD", "eval")
# Side work of the transformers up to this width alternates between TWO side streams (GPT._offload_side): their blocks are
# launch-bound on the chain AND on the side stream (8 against ~14 launches of 5-8 us per block), so the join at the end of the
# transformer waited for the side stream.  Measured (3 interleaved runs each): bf16 step 17.55 -> 17.16 ms, fp32 31.62 -> 31.37;
# C = 512 included: 17.33 / 31.50 (its side work is chip-filling GEMMs); three or four streams: 17.57 / 18.02 and 31.95 (every
# further fork is a queue hop).
SIDE_SPLIT_MAX_C = 256
# The narrow transformers (n_embd 64 / 128, fp32): a block is 2 forward launches (ln1 + q/k/v + attention per (sample, head, half);
# proj + ln2 + mlp per 32-row block) and 3 backward launches (dQ, dK/dV, the row-local rest) instead of 8 and 9 on the dependent
# chain (ops.gpt_block_*, csrc/gpt_block.hip).  MMFN_GPT_FUSED: "1" (default) both passes, "fwd" forward only, "0" off.
GPT_FUSED = os.environ.get("MMFN_GPT_FUSED", "1")


def _dw_db(dy, x, gw, gb):
    """A Linear's weight AND bias gradient: one GEMM in fp32 (the bias gradient = the column sums of dy, formed from the operand
    fragments of dW = dy^T x: MMFN_EPI_COLSUM_A), column sums + GEMM in the bf16 mode."""
    if dy.dtype == torch.bfloat16:
        ops.colsum(dy, gb)
        return ops.linear_dw(dy, x, out=gw)
    return ops.linear_dw(dy, x, out=gw, db=gb)


class GPT(object):
    """model_vec.py:136-246 (GPT), :112-133 (Block), :73-109 (SelfAttention)."""

    def __init__(self, name, layout, prefix, mod, cfg, stream_base):
        self.name, self.mod = name, mod
        self.C = mod.n_embd
        self.nh = cfg.n_head
        self.hs = self.C // self.nh
        self.T = mod.pos_emb.shape[1]
        self.n_modal = mod.n_modal
        self.pos, self.g_pos = layout.w(prefix + ".pos_emb"), layout.g(prefix + ".pos_emb")
        self.vel = Linear(layout, prefix + ".vel_emb")
        self.stream_base = stream_base
        self.blocks = []
        C = self.C
        for i in range(len(mod.blocks)):
            bp = "%s.blocks.%d" % (prefix, i)
            blk = {
                "ln1": LayerNorm("%s.b%d.ln1" % (name, i), layout, bp + ".ln1"),
                "ln2": LayerNorm("%s.b%d.ln2" % (name, i), layout, bp + ".ln2"),
                "proj": Linear(layout, bp + ".attn.proj"),
                "fc1": Linear(layout, bp + ".mlp.0"),
                "fc2": Linear(layout, bp + ".mlp.2"),
            }
            blk["wqkv"], blk["g_wqkv"] = layout.packed(bp + ".attn.key.weight", 3 * C, C)
            blk["wqkv_name"] = bp + ".attn.key.weight"
            blk["wqkv16"] = blk["wqkv16t"] = None   # bf16 shadows (Engine._build_shadows)
            blk["bqkv"], blk["g_bqkv"] = layout.packed(bp + ".attn.key.bias", 3 * C)
            blk["fold"] = None   # (wqkv * gamma1, c1, c2, fc1.w * gamma2, c1, c2): Engine._build_ln_fold
            self.blocks.append(blk)
        self.ln_f = LayerNorm(name + ".ln_f", layout, prefix + ".ln_f")
        self._fin_tables = {}   # device pointer tables of the batched LayerNorm finalize, per buffer set

    def fwd(self, ctx, feats, velocity):
        B = velocity.shape[0]
        T, C, nh, hs = self.T, self.C, self.nh, self.hs
        M = B * T
        bufs, nm = ctx.bufs, self.name
        p_embd, p_attn, p_resid = ctx.drop
        adt = ctx.adt
        # The residual stream x (and, in bwd, its gradient) is fp32 in BOTH modes: in the bf16 mode the LayerNorm outputs, q / k / v,
        # the attention output and the MLP hidden - every GEMM operand - are bf16, the sums x + proj(.) and x + mlp(.) are not
        # (what torch.autocast does to model_vec.py:124-132; rounding the stream alone costs 0.03-0.04 of gradient cosine per
        # backward stage, tools/experiments/bf16_where.py)
        sdt = torch.float32
        Wf = (lambda lin: lin.w16) if ctx.bf16 else (lambda lin: lin.w)      # forward operand of a Linear
        x = bufs.get(nm + ".x0", (B, T, C), sdt)
        ops.tokens_fwd(feats, self.pos.view(T, C), self.vel.w.view(C), self.vel.b, velocity, x, p_embd, ctx.rng_state,
                       self.stream_base, frames=self.frames)
        self.velocity = velocity
        x = x.view(M, C)
        self.acts = []
        scale = 1.0 / math.sqrt(hs)
        nb = len(self.blocks)
        # the activations the weight gradients read again, one slab per block (the side stream may lag by several blocks)
        S_a, S_a2 = bufs.get(nm + ".S.a", (nb, M, C), adt), bufs.get(nm + ".S.a2", (nb, M, C), adt)
        S_o, S_h = bufs.get(nm + ".S.att", (nb, M, C), adt), bufs.get(nm + ".S.h", (nb, M, 4 * C), adt)
        self.stacks = (S_a, S_a2, S_o, S_h)
        fold = (not ctx.bf16) and ops.current_precision() == "f32" and self.blocks[0]["fold"] is not None and M % 64 == 0 \
            and ctx.engine.ln_fold_now(ctx.training)
        self.folded_fwd = fold
        fused = self.fused_now(ctx) and not fold          # fp32: both fused launches; bf16 mode: the row-block launch after attention16
        self.fused_fwd = fused
        for i, blk in enumerate(self.blocks):
            sb = self.stream_base + 1 + 3 * i
            qkv = bufs.get("%s.b%d.qkv" % (nm, i), (M, 3 * C), adt)
            if fused:
                ln1, ln2 = blk["ln1"], blk["ln2"]
                mu1, rs1 = bufs.get(ln1.name + ".mu", (M,)), bufs.get(ln1.name + ".rs", (M,))
                mu2, rs2 = bufs.get(ln2.name + ".mu", (M,)), bufs.get(ln2.name + ".rs", (M,))
                lse = bufs.get("%s.b%d.lse" % (nm, i), (B, nh, T))
                x1 = bufs.get("%s.b%d.x1" % (nm, i), (M, C), sdt)
                x2 = bufs.get("%s.b%d.x2" % (nm, i), (M, C), sdt)
                a, o, a2, h = S_a[i], S_o[i], S_a2[i], S_h[i]
                if not self.fused_attn_now(ctx):   # bf16 mode, or a token count the fused attention launch is not shaped for (rad: 256)
                    ln1.fwd(ctx, x, out=a)
                    ops.linear_fwd(a, blk["wqkv16"] if ctx.bf16 else blk["wqkv"], blk["bqkv"], out=qkv)
                    ops.attention_fwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, C, lse, B, T, nh, hs, scale, drop_p=p_attn,
                                      rng_state=ctx.rng_state, rng_stream=sb)
                    d = self._desc(blk, B, ctx, sb, weights="fwd16" if ctx.bf16 else "f32", x=x, o=o, x1=x1, a2=a2, mu2=mu2, rs2=rs2,
                                   h=h, x2=x2)
                else:
                    d = self._desc(blk, B, ctx, sb, x=x, a=a, mu1=mu1, rs1=rs1, qkv=qkv, o=o, lse=lse, x1=x1, a2=a2, mu2=mu2,
                                   rs2=rs2, h=h, x2=x2)
                    ops.gpt_block_attn_fwd(d)
                    ln1.saved = (x, mu1, rs1, ACT_NONE)
                ops.gpt_block_mlp_fwd(d)
                ln2.saved = (x1, mu2, rs2, ACT_NONE)
                self.acts.append((x, a, qkv, o, lse, x1, a2, h))
                x = x2
                continue
            if fold:
                # LN(x) . Wqkv^T + b in one launch; a = LN(x) is (re)computed by the backward's side work, where it is needed
                ln = blk["ln1"]
                mu, rs = bufs.get(ln.name + ".mu", (M,)), bufs.get(ln.name + ".rs", (M,))
                wf, c1, c2 = blk["fold"][:3]
                ops.linear_fwd(x, wf, c2, out=qkv, ln_fold=(c1, mu, rs, 1e-5))
                ln.saved = (x, mu, rs, ACT_NONE)
                a = S_a[i]
            else:
                a = blk["ln1"].fwd(ctx, x, out=S_a[i])
                ops.linear_fwd(a, blk["wqkv16"] if ctx.bf16 else blk["wqkv"], blk["bqkv"], out=qkv)
            o = S_o[i]
            lse = bufs.get("%s.b%d.lse" % (nm, i), (B, nh, T))
            # packed columns: [key | query | value]  (reference registration order, model_vec.py:82-84)
            ops.attention_fwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, C, lse, B, T, nh, hs, scale, drop_p=p_attn,
                              rng_state=ctx.rng_state, rng_stream=sb)
            x1 = bufs.get("%s.b%d.x1" % (nm, i), (M, C), sdt)
            ops.linear_fwd(o, Wf(blk["proj"]), blk["proj"].b, out=x1, res=x, ldr=C, drop_p=p_resid, rng_state=ctx.rng_state,
                           rng_stream=sb + 1)
            h = S_h[i]
            if fold:
                ln = blk["ln2"]
                mu, rs = bufs.get(ln.name + ".mu", (M,)), bufs.get(ln.name + ".rs", (M,))
                wf, c1, c2 = blk["fold"][3:]
                ops.linear_fwd(x1, wf, c2, out=h, relu=True, ln_fold=(c1, mu, rs, 1e-5))
                ln.saved = (x1, mu, rs, ACT_NONE)
                a2 = S_a2[i]
            else:
                a2 = blk["ln2"].fwd(ctx, x1, out=S_a2[i])
                ops.linear_fwd(a2, Wf(blk["fc1"]), blk["fc1"].b, out=h, relu=True)
            x2 = bufs.get("%s.b%d.x2" % (nm, i), (M, C), sdt)
            ops.linear_fwd(h, Wf(blk["fc2"]), blk["fc2"].b, out=x2, res=x1, ldr=C, drop_p=p_resid, rng_state=ctx.rng_state,
                           rng_stream=sb + 2)
            self.acts.append((x, a, qkv, o, lse, x1, a2, h))
            x = x2
        y = self.ln_f.fwd(ctx, x, out=bufs.get(nm + ".ln_f.out", (M, C), adt))
        return y.view(B, T, C)

    def fused_now(self, ctx):
        """The fused block kernels serve this transformer in this pass: fp32 mode and arithmetic, n_embd 64 / 128, 4 heads, T = 192."""
        return (GPT_FUSED != "0" and (ctx.bf16 or ops.current_precision() == "f32")
                and ops.gpt_block_rows_supported(self.C, self.T))

    def fused_attn_now(self, ctx):
        """... and the attention launch with the projections as its prologue too (fp32, 4 heads, 192 tokens)."""
        return not ctx.bf16 and ops.gpt_block_supported(self.C, self.nh, self.T)

    def _desc(self, blk, B, ctx, sb, sb_below=0, below_colsum=False, weights="f32", **tensors):
        """weights: "f32" the master weights; "fwd16" / "bwd16" the bf16 mode's [out][in] / transposed [in][out] shadows."""
        p_embd, p_attn, p_resid = ctx.drop
        if weights == "f32":
            w = dict(wqkv=blk["wqkv"], wproj=blk["proj"].w, w1=blk["fc1"].w, w2=blk["fc2"].w)
        elif weights == "fwd16":
            w = dict(wqkv=blk["wqkv16"], wproj=blk["proj"].w16, w1=blk["fc1"].w16, w2=blk["fc2"].w16)
        else:
            w = dict(wqkv=blk["wqkv16t"], wproj=blk["proj"].w16t, w1=blk["fc1"].w16t, w2=blk["fc2"].w16t)
        return ops.gpt_block_desc(
            B, self.T, self.C, self.nh, eps=1e-5, attn_pdrop=p_attn, resid_pdrop=p_resid, rng_state=ctx.rng_state, rng_stream=sb,
            rng_stream_below=sb_below, below_colsum=below_colsum,
            ln1_w=blk["ln1"].w, ln1_b=blk["ln1"].b, bqkv=blk["bqkv"], bproj=blk["proj"].b,
            ln2_w=blk["ln2"].w, ln2_b=blk["ln2"].b, b1=blk["fc1"].b, b2=blk["fc2"].b, **w, **tensors)

    def _bwd_fused(self, ctx, g_y):
        """GPT.bwd with the row-local chain of every block in one launch (csrc/gpt_block.hip gpt_bwd_rows_kernel): per block the chain is
        dQ -> dK/dV -> rows(this block's qkv dgrad + ln1 backward | the block below's mlp / ln2 / proj dgrads)."""
        B = g_y.shape[0]
        T, C, nh, hs = self.T, self.C, self.nh, self.hs
        M = B * T
        bufs, nm = ctx.bufs, self.name
        p_embd, p_attn, p_resid = ctx.drop
        scale = 1.0 / math.sqrt(hs)
        nblk = len(self.blocks)
        # bf16 mode: the "dropped copy" is also where the stream gradient becomes a bf16 GEMM operand, so it exists without dropout too
        drop = p_resid > 0.0 or ctx.bf16
        sb_of = lambda i: self.stream_base + 1 + 3 * i
        f32, adt = torch.float32, ctx.adt   # the residual stream's gradient (G, G1) is fp32 in both modes
        G = bufs.get(nm + ".S.g", (nblk, M, C), f32)
        GD = bufs.get(nm + ".S.gdrop", (nblk, M, C), adt) if drop else None
        G1 = bufs.get(nm + ".S.g1", (nblk, M, C), f32)
        GD2 = bufs.get(nm + ".S.gdrop2", (nblk, M, C), adt) if drop else None
        GH = bufs.get(nm + ".S.gh", (nblk, M, 4 * C), adt)
        DQKV = bufs.get(nm + ".S.dqkv", (nblk, M, 3 * C), adt)
        nrow = M // ops.GPT_ROWS
        P1 = bufs.get(nm + ".S.part1", (nblk, nrow, 3, C), f32)
        P2 = bufs.get(nm + ".S.part2", (nblk, nrow, 3, C), f32)
        go = bufs.get(nm + ".go", (M, C), adt)
        g_tok = bufs.get(nm + ".g_tok", (M, C), f32)
        delta = bufs.get(nm + ".delta", (B, nh, T))
        side = []
        self.ln_f.bwd(ctx, g_y.view(M, C), out=G[nblk - 1], dropped=GD[nblk - 1] if drop else None, drop_p=p_resid,
                      rng_stream=sb_of(nblk - 1) + 2, colsum=self.blocks[nblk - 1]["fc2"].gb, defer=side)
        descs = []
        for i, blk in enumerate(self.blocks):
            x, a, qkv, o, lse, x1, a2, h = self.acts[i]
            descs.append(self._desc(
                blk, B, ctx, sb_of(i), sb_below=sb_of(i - 1) if i > 0 else 0, below_colsum=i > 0,
                weights="bwd16" if ctx.bf16 else "f32", x=x, mu1=blk["ln1"].saved[1], rs1=blk["ln1"].saved[2], x1=x1, mu2=blk["ln2"].saved[1], rs2=blk["ln2"].saved[2], h=h,
                g=G[i], gd=GD[i] if drop else None, gh=GH[i], g1=G1[i], gd2=GD2[i] if drop else None, go=go, dqkv=DQKV[i],
                g_below=G[i - 1] if i > 0 else g_tok, gd_below=GD[i - 1] if (drop and i > 0) else None, part_ln1=P1[i], part_ln2=P2[i]))
        ops.gpt_block_bwd_rows(None, descs[nblk - 1])
        pending = None
        for i in range(nblk - 1, -1, -1):
            blk = self.blocks[i]
            x, a, qkv, o, lse, x1, a2, h = self.acts[i]
            dqkv = DQKV[i]
            ops.attention_bwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, go, C, lse, delta, dqkv[:, C:], dqkv, dqkv[:, 2 * C:],
                              3 * C, B, T, nh, hs, scale, drop_p=p_attn, rng_state=ctx.rng_state, rng_stream=sb_of(i))
            if pending is not None:   # the block above's side work, now that this block's first chain kernel is captured
                self._offload_side(ctx, pending)
                pending = None
            ops.gpt_block_bwd_rows(descs[i], descs[i - 1] if i > 0 else None)
            gp, gp2 = (GD[i] if drop else G[i]), (GD2[i] if drop else G1[i])
            below_gb = self.blocks[i - 1]["fc2"].gb if i > 0 else None
            side += [
                lambda gp=gp, blk=blk, h=h: ops.linear_dw(gp, h, out=blk["fc2"].gw),
                lambda gh=GH[i], blk=blk, a2=a2: _dw_db(gh, a2, blk["fc1"].gw, blk["fc1"].gb),
                lambda p=P2[i], blk=blk: ops.layernorm_bwd_finalize(p, nrow, C, blk["ln2"].gw, blk["ln2"].gb, blk["proj"].gb),
                lambda gp2=gp2, blk=blk, o=o: ops.linear_dw(gp2, o, out=blk["proj"].gw),
                lambda dqkv=dqkv, blk=blk, a=a: _dw_db(dqkv, a, blk["g_wqkv"], blk["g_bqkv"]),
                lambda p=(P1[i] if i > 0 else P1[i].view(-1)[:nrow * 2 * C].view(nrow, 2, C)), blk=blk, cs=below_gb:
                    ops.layernorm_bwd_finalize(p, nrow, C, blk["ln1"].gw, blk["ln1"].gb, cs),
            ]
            work, side = side, []
            pending = (ctx.fork_point(), work)
        if pending is not None:
            self._offload_side(ctx, pending)
        ctx.rejoin()
        gtok = g_tok.view(B, T, C)
        ops.tokens_bwd(gtok, self.velocity, self.g_pos.view(T, C), self.vel.gw.view(C), self.vel.gb, p_embd, ctx.rng_state,
                       self.stream_base)
        return gtok

    def _offload_side(self, ctx, pending):
        """A block's side work (fork point, launch closures).  The narrow transformers are launch-bound on BOTH streams - per block
        8 kernels on the chain and ~14 on the side stream, 5-8 us each - so there the closures alternate between two side streams."""
        ev, work = pending
        if self.C <= SIDE_SPLIT_MAX_C and ctx.side2 is not None and len(work) > 1:
            ctx.offload_at(ev, lambda: [f() for f in work[0::2]])
            ctx.offload_at(ev, lambda: [f() for f in work[1::2]], second=True)
        else:
            ctx.offload_at(ev, lambda: [f() for f in work])

    def bwd(self, ctx, g_y):
        """g_y: [B,T,C] gradient of the GPT output.  Returns the token gradient (masked by the
        embedding dropout) to be spread back over the feature maps; writes all parameter grads."""
        if GPT_FUSED == "1" and self.fused_now(ctx):
            return self._bwd_fused(ctx, g_y)
        B = g_y.shape[0]
        T, C, nh, hs = self.T, self.C, self.nh, self.hs
        M = B * T
        bufs, nm = ctx.bufs, self.name
        p_embd, p_attn, p_resid = ctx.drop
        scale = 1.0 / math.sqrt(hs)
        nblk = len(self.blocks)
        drop = p_resid > 0.0
        # the LayerNorm backward that produces a block's incoming gradient also writes its dropped copy (the residual
        # dropouts of the forward sit in GEMM epilogues; their masks are re-applied here without an extra pass)
        sb_of = lambda i: self.stream_base + 1 + 3 * i
        # Per-block gradient tensors, stacked over the blocks: G[i] = gradient arriving at block i's output, GD[i] its dropped copy
        # (what enters the MLP branch), G1 / GD2 the same for the attention branch, GH / DQKV the gradients of the hidden / qkv
        # activations.  The side stream reads them for the weight gradients, so they are per block and it may lag by any number
        # of blocks: one rejoin at the end of the transformer.
        adt, sdt = ctx.adt, torch.float32   # sdt: the residual stream's gradient is fp32 in both modes (see fwd)
        # bf16 mode: the "dropped copy" is also where the stream gradient becomes a bf16 GEMM operand, so it exists without dropout too
        drop = drop or ctx.bf16
        Wb = (lambda lin: lin.w16t) if ctx.bf16 else (lambda lin: lin.w)     # data-gradient operand of a Linear (bf16: the transposed shadow)
        G = bufs.get(nm + ".S.g", (nblk, M, C), sdt)
        GD = bufs.get(nm + ".S.gdrop", (nblk, M, C), adt) if drop else None
        G1 = bufs.get(nm + ".S.g1", (nblk, M, C), sdt)
        GD2 = bufs.get(nm + ".S.gdrop2", (nblk, M, C), adt) if drop else None
        GH = bufs.get(nm + ".S.gh", (nblk, M, 4 * C), adt)
        DQKV = bufs.get(nm + ".S.dqkv", (nblk, M, 3 * C), adt)
        # Side work - everything that only feeds the optimizer: weight / bias gradients, column sums, the LayerNorm row
        # reductions - is collected per block and forked to the side stream ONCE per block, and the fork is captured AFTER the
        # chain's next kernel (Ctx.offload_at): in a replayed graph every fork moves the continuation of the chain to another
        # hardware queue, and each such hop costs 10-16 us of idle time (profiles/r02c_graph_timeline.txt) - with a fork per
        # weight gradient and per LayerNorm reduction (7 per block) that was 0.3 ms per transformer.
        side = []
        pending = None
        # bf16 mode: the 17 LayerNorm backward passes leave their row reductions (weight / bias gradient, the consuming Linear's bias
        # gradient) to ONE launch at the end of this transformer (mmfn_layernorm_bwd_finalize_batched_f32) instead of one each on the
        # side streams.  Bit-identical either way; measured (3 interleaved runs each): bf16 step 17.14 -> 17.09 ms, fp32 31.40 -> 31.49
        # (there the launch at the tail costs more than the side streams gain) - hence by mode.
        fins = [] if (ctx.bf16 and ctx.side is not None and DEFER_LN_REDUCTIONS) else None
        refold = getattr(self, "folded_fwd", False)   # the forward ran ln1 / ln2 inside the QKV / mlp.0 GEMMs (LN_FOLD)
        # (one scratch pair per closure site and block parity: consecutive blocks' side work alternates between TWO side streams for
        # C <= SIDE_SPLIT_MAX_C, so a shared pair would be written concurrently)
        scr = [[(bufs.get("%s.ln.scratch.mu%d%d" % (nm, site, par), (M,)), bufs.get("%s.ln.scratch.rs%d%d" % (nm, site, par), (M,)))
                for par in range(2)] for site in range(2)] if refold else None
        g = self.ln_f.bwd(ctx, g_y.view(M, C), out=G[nblk - 1], dropped=GD[nblk - 1] if drop else None, drop_p=p_resid,
                          rng_stream=sb_of(nblk - 1) + 2, colsum=self.blocks[nblk - 1]["fc2"].gb, defer=side, collect=fins)
        for i in range(nblk - 1, -1, -1):
            blk = self.blocks[i]
            sb = sb_of(i)
            x, a, qkv, o, lse, x1, a2, h = self.acts[i]
            # ---- MLP branch: x2 = x1 + drop(fc2(relu(fc1(ln2(x1)))))
            gp = GD[i] if drop else g
            side.append(lambda gp=gp, blk=blk, h=h: ops.linear_dw(gp, h, out=blk["fc2"].gw))   # fc2.gb: from the LayerNorm backward
            gh = GH[i]
            ghpart = None
            if ctx.bf16:
                # the column sums of gh (= mlp.0's bias gradient) come out of this GEMM's epilogue as partial rows
                from . import ops16
                ghpart = bufs.get("%s.b%d.ghpart" % (nm, i), (ops16.max_stats_rows(M), 2, 4 * C), torch.float64)
                ops.linear_dx(gp, Wb(blk["fc2"]), out=gh, aux=h, ldaux=4 * C, stats=ghpart, stats_mode=1)
                ghrows = ops16.gemm_stats_rows(ops16.G16_NT, M, 4 * C, C)
            else:
                ops.linear_dx(gp, Wb(blk["fc2"]), out=gh, aux=h, ldaux=4 * C)
            if pending is not None:   # the previous block's side work, now that this block's first chain kernel is captured
                self._offload_side(ctx, pending)
                pending = None
            if ghpart is not None:
                side.append(lambda gh=gh, blk=blk, a2=a2, p=ghpart, r=ghrows: (ops16.colsum_partials(p, r, 4 * C, blk["fc1"].gb),
                                                                               ops.linear_dw(gh, a2, out=blk["fc1"].gw)))
            elif refold:
                # the forward folded ln2 into mlp.0's GEMM: the normalised tensor the weight gradient contracts with is made here
                side.append(lambda gh=gh, blk=blk, a2=a2, x1=x1, sc=scr[0][i & 1]: (ops.layernorm_fwd(x1, blk["ln2"].w, blk["ln2"].b, a2, sc[0], sc[1]),
                                                                  _dw_db(gh, a2, blk["fc1"].gw, blk["fc1"].gb)))
            else:
                side.append(lambda gh=gh, blk=blk, a2=a2: (_dw_db(gh, a2, blk["fc1"].gw, blk["fc1"].gb)))
            ga2 = bufs.get(nm + ".ga", (M, C), adt)
            ops.linear_dx(gh, Wb(blk["fc1"]), out=ga2)
            g1 = blk["ln2"].bwd(ctx, ga2, dres=g, out=G1[i], dropped=GD2[i] if drop else None, drop_p=p_resid,
                                rng_stream=sb + 1, colsum=blk["proj"].gb, defer=side, collect=fins)
            # ---- attention branch: x1 = x + drop(proj(att(ln1(x))))
            gp = GD2[i] if drop else g1
            side.append(lambda gp=gp, blk=blk, o=o: ops.linear_dw(gp, o, out=blk["proj"].gw))   # proj.gb: from ln2's backward
            go = bufs.get(nm + ".go", (M, C), adt)
            ops.linear_dx(gp, Wb(blk["proj"]), out=go)
            dqkv = DQKV[i]
            delta = bufs.get(nm + ".delta", (B, nh, T))
            ops.attention_bwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, go, C, lse, delta, dqkv[:, C:], dqkv, dqkv[:, 2 * C:],
                              3 * C, B, T, nh, hs, scale, drop_p=p_attn, rng_state=ctx.rng_state, rng_stream=sb)
            if refold:
                side.append(lambda dqkv=dqkv, blk=blk, a=a, x=x, sc=scr[1][i & 1]: (ops.layernorm_fwd(x, blk["ln1"].w, blk["ln1"].b, a, sc[0], sc[1]),
                                                                  _dw_db(dqkv, a, blk["g_wqkv"], blk["g_bqkv"])))
            else:
                side.append(lambda dqkv=dqkv, blk=blk, a=a: (_dw_db(dqkv, a, blk["g_wqkv"], blk["g_bqkv"])))
            ga = bufs.get(nm + ".ga2", (M, C), adt)
            ops.linear_dx(dqkv, blk["wqkv16t"] if ctx.bf16 else blk["wqkv"], out=ga)
            g = blk["ln1"].bwd(ctx, ga, dres=g1, out=G[i - 1] if i > 0 else bufs.get(nm + ".g_tok", (M, C), sdt),
                               dropped=GD[i - 1] if (drop and i > 0) else None, drop_p=p_resid,
                               rng_stream=sb_of(i - 1) + 2, colsum=self.blocks[i - 1]["fc2"].gb if i > 0 else None, defer=side,
                               collect=fins)
            work, side = side, []
            pending = (ctx.fork_point(), work)   # enqueued after the next block's first kernel (Ctx.offload_at)
        if pending is not None:
            self._offload_side(ctx, pending)
        if fins:
            # keyed by the addresses the table holds (not by id(bufs): a dropped Buffers object's id can be reused, and a
            # reallocated gradient buffer moves the targets) - a stale table would write through dangling pointers
            key = tuple(0 if t is None else t.data_ptr() for f in fins for t in f[:4])
            tab = self._fin_tables.get(key)
            if tab is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("LayerNorm finalize table of %s first built during graph capture; run one eager step first" % nm)
                tab = self._fin_tables[key] = ops.layernorm_finalize_table([f[:4] for f in fins], g.device)
            ctx.offload(lambda: ops.layernorm_bwd_finalize_batched(tab, len(fins), fins[0][4], fins[0][5]))
        ctx.rejoin()
        gtok = g.view(B, T, C)
        ops.tokens_bwd(gtok, self.velocity, self.g_pos.view(T, C), self.vel.gw.view(C), self.vel.gb, p_embd, ctx.rng_state,
                       self.stream_base)
        return gtok


# ----------------------------------------------------------------------------- VectorNet
class VectorNet(object):
    """model_vec.py:326-416.  Only lane 0's fused token feeds the generator (:412), so everything
    after the lane attention runs on B rows; the constant pos_emb branch (fed zeros, :408) runs on
    a single row."""

    def __init__(self, name, layout, prefix, mod):
        self.name = name
        self.sub = []
        for i in range(3):
            p = "%s.lane_subgraph.layers.mlp_%d.mlp" % (prefix, i)
            self.sub.append((Linear(layout, p + ".0"), LayerNorm("%s.sub%d.ln" % (name, i), layout, p + ".1")))
        self.pe0, self.pe_ln, self.pe3 = (Linear(layout, prefix + ".pos_emb.0"), LayerNorm(name + ".pe.ln", layout, prefix + ".pos_emb.1"),
                                          Linear(layout, prefix + ".pos_emb.3"))
        self.qkv = Linear(layout, prefix + ".L2L.to_qkv", bias=False)
        self.to_out = Linear(layout, prefix + ".L2L.to_out.0")
        self.af0, self.af_ln, self.af3 = (Linear(layout, prefix + ".agent_fusion.0"), LayerNorm(name + ".af.ln", layout, prefix + ".agent_fusion.1"),
                                          Linear(layout, prefix + ".agent_fusion.3"))
        self.gen0, self.gen_ln, self.gen3 = (Linear(layout, prefix + ".generator.0"), LayerNorm(name + ".gen.ln", layout, prefix + ".generator.1"),
                                             Linear(layout, prefix + ".generator.3"))
        self.heads = mod.L2L.heads

    def fwd(self, ctx, lane, lane_num):
        """lane [B,L,n,5] f32 nodes (reference format) or pre-vectorised polylines [B,L,V,lane_channels] (perf-only
        64x19x8 variant, config.lane_channels = 8), lane_num int32 [B]  ->  map features NHWC [B,64,64,64]."""
        bufs, nm = ctx.bufs, self.name
        B, L, n, F = lane.shape
        cin = self.sub[0][0].w.shape[1]
        if F == 5 and cin == 7:
            V = n - 1
            R = B * L
            x = ops.lane_to_vector(lane, bufs.get(nm + ".vec", (R * V, 7)))
        elif F == cin:
            V, R = n, B * L
            x = lane.view(R * V, cin)
        else:
            raise ValueError("lane tensor [..., %d, %d] does not fit a VectorNet with lane_channels=%d" % (n, F, cin))
        self.sub_saved = []
        for i, (lin, ln) in enumerate(self.sub):
            pre = bufs.get("%s.sub%d.pre" % (nm, i), (R * V, 64))
            ops.linear_fwd(x, lin.w, lin.b, out=pre)
            y = ln.fwd(ctx, pre, ACT_RELU)
            last = i == len(self.sub) - 1
            out = bufs.get("%s.sub%d.cat" % (nm, i), (R, 128) if last else (R * V, 128))
            arg = bufs.get("%s.sub%d.arg" % (nm, i), (R, 64), torch.uint8)
            ops.polyline_pool_fwd(y, out, arg, R, V, 64, last)
            self.sub_saved.append((x, arg))
            x = out
        tok = x  # [R,128] lane tokens
        qkv = bufs.get(nm + ".qkv", (R, 384))
        ops.linear_fwd(tok, self.qkv.w, None, out=qkv)
        # Only lane 0's attended token feeds the encoder output (model_vec.py:412), so the lane attention runs for query 0
        # alone: O(L) instead of O(L^2), and any number of lanes (the fused attention kernels stop at 256 tokens)
        hd = 128 // self.heads
        att0 = bufs.get(nm + ".att0", (B, 128))
        prob = bufs.get(nm + ".prob", (B, self.heads, L))
        ops.lane0_attention_fwd(qkv, lane_num, B, L, self.heads, hd, hd ** -0.5, att0, prob)
        t0 = bufs.get(nm + ".t0", (B, 128))
        ops.gemm(att0, self.to_out.w, t0, B, 128, 128, 128, 128, 128, bias=self.to_out.b)
        # constant positional branch: pos_emb(zeros) = Linear(GELU(LN(bias0)))
        pe_pre = bufs.get(nm + ".pe.pre", (1, 64))
        zero2 = bufs.get(nm + ".zero2", (1, 2))
        ops.fill(zero2, 0.0)
        ops.linear_fwd(zero2, self.pe0.w, self.pe0.b, out=pe_pre)
        pe_act = self.pe_ln.fwd(ctx, pe_pre, ACT_GELU)
        pe = bufs.get(nm + ".pe", (1, 64))
        ops.linear_fwd(pe_act, self.pe3.w, self.pe3.b, out=pe)
        # agent_fusion.0 over cat([t0, pe]): split the [128,192] weight into its two column blocks
        cbias = bufs.get(nm + ".af.cbias", (1, 128))
        ops.gemm(pe, self.af0.w[:, 128:], cbias, 1, 128, 64, 64, 192, 128, bias=self.af0.b)
        af_pre = bufs.get(nm + ".af.pre", (B, 128))
        ops.gemm(t0, self.af0.w, af_pre, B, 128, 128, 128, 192, 128, bias=cbias.view(128))
        af_act = self.af_ln.fwd(ctx, af_pre, ACT_GELU)
        fused = bufs.get(nm + ".fused", (B, 128))
        ops.linear_fwd(af_act, self.af3.w, self.af3.b, out=fused)
        gen_pre = bufs.get(nm + ".gen.pre", (B, 64))
        ops.linear_fwd(fused, self.gen0.w, self.gen0.b, out=gen_pre)
        gen_act = self.gen_ln.fwd(ctx, gen_pre, ACT_GELU)
        nchw = bufs.get(nm + ".nchw", (B, 64 * 64 * 64))
        ops.linear_fwd(gen_act, self.gen3.w, self.gen3.b, out=nchw)
        out = bufs.get(nm + ".out", (B, 64, 64, 64), ctx.adt)   # (bf16 mode: VectorNet is fp32 inside, its map feature is an activation)
        ops.transpose(nchw, out, B, 64, 4096)  # "b (n d a)" -> NHWC [b, d, a, n]
        self.saved = (B, L, V, tok, qkv, att0, prob, lane_num, t0, pe, pe_act, af_act, fused, gen_act)
        return out

    def bwd(self, ctx, g_out):
        bufs, nm = ctx.bufs, self.name
        B, L, V, tok, qkv, att0, prob, lane_num, t0, pe, pe_act, af_act, fused, gen_act = self.saved
        R = B * L
        g_nchw = bufs.get(nm + ".g.nchw", (B, 64 * 64 * 64))
        ops.transpose(g_out.view(B, 4096, 64), g_nchw, B, 4096, 64)
        ops.colsum(g_nchw, self.gen3.gb)
        ops.linear_dw(g_nchw, gen_act, out=self.gen3.gw)
        g_gen_act = ops.linear_dx(g_nchw, self.gen3.w, out=bufs.get(nm + ".g.gen_act", (B, 64)))
        g_gen_pre = self.gen_ln.bwd(ctx, g_gen_act)
        ops.colsum(g_gen_pre, self.gen0.gb)
        ops.linear_dw(g_gen_pre, fused, out=self.gen0.gw)
        g_fused = ops.linear_dx(g_gen_pre, self.gen0.w, out=bufs.get(nm + ".g.fused", (B, 128)))
        ops.colsum(g_fused, self.af3.gb)
        ops.linear_dw(g_fused, af_act, out=self.af3.gw)
        g_af_act = ops.linear_dx(g_fused, self.af3.w, out=bufs.get(nm + ".g.af_act", (B, 128)))
        g_af_pre = self.af_ln.bwd(ctx, g_af_act)
        # agent_fusion.0: columns 0:128 act on t0, columns 128:192 on the constant pe row
        gsum = ops.colsum(g_af_pre, bufs.get(nm + ".g.afsum", (1, 128)).view(128)).view(1, 128)
        ops.axpby(self.af0.gb, gsum.view(128), 1.0, 0.0)
        ops.gemm(g_af_pre, t0, self.af0.gw, 128, 128, B, 128, 128, 192, a_mode=ops.A_COLMAJOR, b_mode=ops.B_KN)
        ops.gemm(gsum, pe, self.af0.gw[:, 128:], 128, 64, 1, 128, 64, 192, a_mode=ops.A_COLMAJOR, b_mode=ops.B_KN)
        g_t0 = bufs.get(nm + ".g.t0", (B, 128))
        ops.gemm(g_af_pre, self.af0.w, g_t0, B, 128, 128, 128, 192, 128, b_mode=ops.B_KN)
        g_pe = bufs.get(nm + ".g.pe", (1, 64))
        ops.gemm(gsum, self.af0.w[:, 128:], g_pe, 1, 64, 128, 128, 192, 64, b_mode=ops.B_KN)
        # pos_emb branch (single row; its first Linear sees a zero input -> exactly-zero weight grad)
        ops.axpby(self.pe3.gb, g_pe.view(64), 1.0, 0.0)
        ops.linear_dw(g_pe, pe_act, out=self.pe3.gw)
        g_pe_act = ops.linear_dx(g_pe, self.pe3.w, out=bufs.get(nm + ".g.pe_act", (1, 64)))
        g_pe_pre = self.pe_ln.bwd(ctx, g_pe_act)
        ops.axpby(self.pe0.gb, g_pe_pre.view(64), 1.0, 0.0)
        ops.fill(self.pe0.gw, 0.0)
        # to_out on lane 0 only
        ops.colsum(g_t0, self.to_out.gb)
        ops.gemm(g_t0, att0, self.to_out.gw, 128, 128, B, 128, 128, 128, a_mode=ops.A_COLMAJOR, b_mode=ops.B_KN)
        g_att0 = bufs.get(nm + ".g.att0", (B, 128))
        ops.gemm(g_t0, self.to_out.w, g_att0, B, 128, 128, 128, 128, 128, b_mode=ops.B_KN)
        dqkv = bufs.get(nm + ".g.qkv", (R, 384))
        hd = 128 // self.heads
        ops.lane0_attention_bwd(qkv, prob, g_att0, lane_num, B, L, self.heads, hd, hd ** -0.5, dqkv)
        ops.linear_dw(dqkv, tok, out=self.qkv.gw)
        g = ops.linear_dx(dqkv, self.qkv.w, out=bufs.get(nm + ".g.tok", (R, 128)))
        for i in range(len(self.sub) - 1, -1, -1):
            lin, ln = self.sub[i]
            x_in, arg = self.sub_saved[i]
            last = i == len(self.sub) - 1
            gy = ops.polyline_pool_bwd(g, arg, bufs.get("%s.g.sub%d.y" % (nm, i), (R * V, 64)), R, V, 64, last)
            g_pre = ln.bwd(ctx, gy)
            ops.colsum(g_pre, lin.gb)
            ops.linear_dw(g_pre, x_in, out=lin.gw)
            if i > 0:
                g = ops.linear_dx(g_pre, lin.w, out=bufs.get("%s.g.sub%d.x" % (nm, i), (R * V, 128)))



# ----------------------------------------------------------------------------- radar GAT
class RadarGAT(object):
    """model_rad.py:853-884 (SpGAT) with its two SpGraphAttentionLayer heads (:778-847).

    radar [B,81,5], adj [B,81,81] -> NHWC map [B,8,8,512] (log-softmax over channels).  Matrix products
    run on the (batched) MFMA GEMM; the (nhid == N == 81) coincidence that makes `Wh @ a` an [81,81]
    score matrix (config.py:53) is kept as is."""

    def __init__(self, name, layout, prefix, mod):
        self.name = name
        self.nh = mod.nheads
        self.alpha = mod.alpha
        self.p = mod.dropout
        self.W = [layout.w("%s.attention_%d.W" % (prefix, i)) for i in range(self.nh)]
        self.gW = [layout.g("%s.attention_%d.W" % (prefix, i)) for i in range(self.nh)]
        self.a = [layout.w("%s.attention_%d.a" % (prefix, i)) for i in range(self.nh)]
        self.ga = [layout.g("%s.attention_%d.a" % (prefix, i)) for i in range(self.nh)]
        self.m1 = Linear(layout, prefix + ".mlp_1.0")
        self.m2 = Linear(layout, prefix + ".mlp_2.0")
        self.stream_base = 900

    def _drop(self, ctx, t, tag, sid):
        p = self.p if ctx.training else 0.0
        if p <= 0.0:
            return t
        return ops.dropout_apply(t, ctx.bufs.get("%s.%s" % (self.name, tag), t.shape), p, ctx.rng_state, self.stream_base + sid)

    def fwd(self, ctx, radar, adj):
        bufs, nm = ctx.bufs, self.name
        B, N, Fin = radar.shape  # 81 nodes, 5 features
        H2 = self.W[0].shape[1]  # 162
        R = B * N
        p = self.p if ctx.training else 0.0
        x = self._drop(ctx, radar.view(R, Fin), "xd", 0)
        cat = bufs.get(nm + ".cat", (B, self.nh * N, H2))
        self.heads = []
        adj2 = adj.view(R, N)
        for h in range(self.nh):
            wh = bufs.get("%s.wh%d" % (nm, h), (R, H2))
            ops.gemm(x, self.W[h], wh, R, H2, Fin, Fin, H2, H2, b_mode=ops.B_KN)
            epre = bufs.get("%s.epre%d" % (nm, h), (R, N))
            ops.gemm(wh, self.a[h], epre, R, N, H2, H2, N, N, b_mode=ops.B_KN)
            prob = bufs.get("%s.p%d" % (nm, h), (R, N))
            att = bufs.get("%s.att%d" % (nm, h), (R, N))
            ops.gat_softmax_fwd(epre, adj2, self.alpha, prob, att, p, ctx.rng_state, self.stream_base + 1 + h)
            # h' = att @ Wh per sample, written into rows [h*81, (h+1)*81) of the concatenated tensor
            ops.gemm(att, wh, cat[:, h * N:], N, H2, N, N, H2, H2, b_mode=ops.B_KN, batch=B, strideA=N * N,
                     strideB=N * H2, strideC=self.nh * N * H2)
            self.heads.append((wh, epre, prob, att))
        n_cat = B * self.nh * N
        y1 = ops.elu_fwd(cat, bufs.get(nm + ".y1", cat.shape))                 # F.elu inside each head
        y1d = self._drop(ctx, y1, "y1d", 4)
        y2 = ops.elu_fwd(y1d, bufs.get(nm + ".y2", cat.shape))                # F.elu before mlp_1
        m1 = bufs.get(nm + ".m1", (n_cat, 256))
        ops.linear_fwd(y2.view(n_cat, H2), self.m1.w, self.m1.b, out=m1)
        m1d = self._drop(ctx, m1, "m1d", 5)
        m1t = bufs.get(nm + ".m1t", (B * 256, self.nh * N))
        ops.transpose(m1d, m1t, B, self.nh * N, 256)
        m2 = bufs.get(nm + ".m2", (B * 256, 128))
        ops.linear_fwd(m1t, self.m2.w, self.m2.b, out=m2)
        m2d = self._drop(ctx, m2, "m2d", 6)
        out = bufs.get(nm + ".out", (B, 8, 8, 512))
        ops.log_softmax_fwd(m2d, out, B * 64, 512, True)
        self.saved = (B, N, Fin, H2, x, adj2, y1, y2, m1t, out)
        if ctx.bf16:   # fp32 inside (81 points x 7 heads: nothing for the bf16 pipe), the radar feature joins the fusion as a bf16 activation
            return ops.cast_to_bf16(out, bufs.get(nm + ".out16", out.shape, ctx.adt))
        return out

    def bwd(self, ctx, g_out):
        bufs, nm = ctx.bufs, self.name
        B, N, Fin, H2, x, adj2, y1, y2, m1t, out = self.saved
        R = B * N
        n_cat = B * self.nh * N
        p = self.p if ctx.training else 0.0
        if g_out.dtype != torch.float32:
            g_out = ops.cast_to_f32(g_out, bufs.get(nm + ".g.out32", g_out.shape))
        g_m2 = ops.log_softmax_bwd(g_out, out, bufs.get(nm + ".g.m2", (B * 256, 128)), B * 64, 512, True)
        if p > 0.0:
            ops.dropout_apply(g_m2, g_m2, p, ctx.rng_state, self.stream_base + 6)
        ops.colsum(g_m2, self.m2.gb)
        ops.linear_dw(g_m2, m1t, out=self.m2.gw)
        g_m1t = ops.linear_dx(g_m2, self.m2.w, out=bufs.get(nm + ".g.m1t", m1t.shape))
        g_m1 = ops.transpose(g_m1t, bufs.get(nm + ".g.m1", (n_cat, 256)), B, 256, self.nh * N)
        if p > 0.0:
            ops.dropout_apply(g_m1, g_m1, p, ctx.rng_state, self.stream_base + 5)
        ops.colsum(g_m1, self.m1.gb)
        ops.linear_dw(g_m1, y2.view(n_cat, H2), out=self.m1.gw)
        g_y2 = ops.linear_dx(g_m1, self.m1.w, out=bufs.get(nm + ".g.y2", (n_cat, H2)))
        g_y1d = ops.elu_bwd(g_y2, y2, bufs.get(nm + ".g.y1d", (n_cat, H2)))
        if p > 0.0:
            ops.dropout_apply(g_y1d, g_y1d, p, ctx.rng_state, self.stream_base + 4)
        g_cat = ops.elu_bwd(g_y1d, y1, bufs.get(nm + ".g.cat", (B, self.nh * N, H2)))
        for h in range(self.nh):
            wh, epre, prob, att = self.heads[h]
            gh = g_cat[:, h * N:]  # [B, 81, 162] view, batch stride nh*N*H2
            # h' = att @ Wh:  d_att = g h'  Wh^T   ;   d_Wh = att^T g h'
            g_att = bufs.get("%s.g.att%d" % (nm, h), (R, N))
            ops.gemm(gh, wh, g_att, N, N, H2, H2, H2, N, b_mode=ops.B_NK, batch=B, strideA=self.nh * N * H2, strideB=N * H2,
                     strideC=N * N)
            g_wh = bufs.get("%s.g.wh%d" % (nm, h), (R, H2))
            ops.gemm(att, gh, g_wh, N, H2, N, N, H2, H2, a_mode=ops.A_COLMAJOR, b_mode=ops.B_KN, batch=B, strideA=N * N,
                     strideB=self.nh * N * H2, strideC=N * H2)
            g_epre = bufs.get("%s.g.epre%d" % (nm, h), (R, N))
            ops.gat_softmax_bwd(g_att, prob, epre, adj2, self.alpha, g_epre, p, ctx.rng_state, self.stream_base + 1 + h)
            # e_pre = Wh @ a
            ops.gemm(wh, g_epre, self.ga[h], H2, N, R, H2, N, N, a_mode=ops.A_COLMAJOR, b_mode=ops.B_KN)
            ops.gemm(g_epre, self.a[h], g_wh, R, H2, N, N, N, H2, b_mode=ops.B_NK, accum=True)
            # Wh = x @ W
            ops.gemm(x, g_wh, self.gW[h], Fin, H2, R, Fin, H2, H2, a_mode=ops.A_COLMAJOR, b_mode=ops.B_KN)


# ----------------------------------------------------------------------------- waypoint head
class Head(object):
    """model_vec.py:642-651,664-680: join MLP 512-256-128-64 (ReLU), GRUCell x pred_len, Linear(64,2)."""

    def __init__(self, layout, pred_len):
        self.join = [Linear(layout, "join.0"), Linear(layout, "join.2"), Linear(layout, "join.4")]
        self.w_ih, self.w_hh = layout.w("decoder.weight_ih"), layout.w("decoder.weight_hh")
        self.b_ih, self.b_hh = layout.w("decoder.bias_ih"), layout.w("decoder.bias_hh")
        self.out = Linear(layout, "output")
        self.grads = [layout.g("decoder.weight_ih"), layout.g("decoder.weight_hh"), layout.g("decoder.bias_ih"),
                      layout.g("decoder.bias_hh"), self.out.gw, self.out.gb]
        self.steps = pred_len

    def fwd(self, ctx, fused, target, gt=None):
        bufs = ctx.bufs
        B = fused.shape[0]
        self.xs = [fused]
        x = fused
        for i, lin in enumerate(self.join):
            y = bufs.get("head.j%d" % i, (B, lin.w.shape[0]))
            ops.linear_fwd(x, lin.w, lin.b, out=y, relu=True)
            self.xs.append(y)
            x = y
        S = self.steps
        pred = bufs.get("head.pred", (B, S, 2))
        train = ctx.training
        hs = bufs.get("head.hs", (B, S + 1, 64)) if train else None
        gates = bufs.get("head.gates", (B, S, 4, 64)) if train else None
        xin = bufs.get("head.xin", (B, S, 2)) if train else None
        lt = bufs.get("head.lt", (B,)) if gt is not None else None
        loss = bufs.get("head.loss", (1,)) if gt is not None else None
        ops.gru_head_fwd(x, target, self.w_ih, self.w_hh, self.b_ih, self.b_hh, self.out.w, self.out.b, gt, pred, hs, gates, xin,
                         lt, loss, S)
        self.saved = (pred, gt, hs, gates, xin)
        return pred, loss

    def bwd(self, ctx, dpred=None, gscale=None):
        bufs = ctx.bufs
        pred, gt, hs, gates, xin = self.saved
        B, S = pred.shape[0], self.steps
        if gscale is None:
            gscale = 1.0 / (B * S * 2)
        npart = ops.lib().mmfn_gru_head_part_floats()
        part = bufs.get("head.part", (B, npart))
        dz = bufs.get("head.dz", (B, 64))
        ops.gru_head_bwd(pred, gt, dpred, gscale, self.w_ih, self.w_hh, self.out.w, hs, gates, xin, dz, part, S)
        tot = ops.colsum(part, bufs.get("head.partsum", (npart,)))
        o = 0
        for gbuf in self.grads:
            n = gbuf.numel()
            ops.axpby(gbuf, tot[o:o + n], 1.0, 0.0)
            o += n
        # z0 = relu(join.4(.)): mask the GRU's gradient, then walk the MLP back; each dX GEMM applies
        # the ReLU mask of the layer below in its epilogue
        g = ops.relu_mask(dz, self.xs[-1], out=bufs.get("head.gz", dz.shape))
        for i in range(len(self.join) - 1, -1, -1):
            lin, x = self.join[i], self.xs[i]
            ops.colsum(g, lin.gb)
            ops.linear_dw(g, x, out=lin.gw)
            gx = bufs.get("head.gx%d" % i, x.shape)
            if i > 0:
                g = ops.linear_dx(g, lin.w, out=gx, aux=x, ldaux=x.shape[1])
            else:
                g = ops.linear_dx(g, lin.w, out=gx)
        return g


# ----------------------------------------------------------------------------- whole network
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _in_precision(fn):
    """Run an Engine method with its plain GEMMs in the engine's arithmetic (ops.precision)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        with ops.precision(self.gemm_dtype):
            return fn(self, *args, **kwargs)
    return wrapped


class Engine(object):
    """Forward / backward / optimizer step of one MMFN replica on one GPU.

    Orchestration restates Encoder.forward (model_vec.py:488-598; model_img.py:310-423) and
    MMFN.forward (model_vec.py:653-682).  seq_len = 1, n_views = 1 is the reference's configuration (config.py:6,11) and the
    tuned one; other values run the same kernels with more frames per sample: a sample's n_views * seq_len camera frames,
    seq_len LiDAR frames and seq_len map frames are batch entries of their branch and token groups of one sequence
    (self.frames / self.group_base; model_img.py:211-246, :410-423)."""

    def __init__(self, module, layout, variant):
        cfg = module.config
        self.module, self.layout, self.variant, self.cfg = module, layout, variant, cfg
        self.device = layout.device
        self.act_dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[getattr(cfg, "act_dtype", "f32")]
        S, V = int(cfg.seq_len), int(cfg.n_views)
        if S < 1 or V < 1:
            raise ValueError("seq_len and n_views must be >= 1, got %d and %d" % (S, V))
        if variant != "img" and S != 1:
            # the reference cannot either: its VectorNet / radar encoders emit one feature map per SAMPLE, which
            # GPT.forward's view(bz, seq_len, ...) rejects for seq_len > 1 (model_vec.py:223-229)
            raise NotImplementedError("seq_len > 1 exists for the image-map model only (the vector-map and radar encoders produce one "
                                      "frame per sample, as in the reference)")
        # frames per sample of each fused modality, in token order, and each modality's first 64-token group
        self.frames = [V * S, S, S]
        self.group_base = [0, V * S, (V + 1) * S]
        tokens = 64 * ((V + 2) * S + (1 if variant == "rad" else 0))
        if tokens > 384:
            raise NotImplementedError("(n_views + 2) * seq_len * 64%s = %d tokens: the attention kernels hold at most 384 keys" % (
                " + 64" if variant == "rad" else "", tokens))
        if self.act_dtype == torch.bfloat16 and tokens > 256:
            raise NotImplementedError("the bf16 training mode's attention kernels stage K and V of one head in LDS: at most 256 tokens, got %d" % tokens)
        if self.device.type != "cuda":
            raise ops._lib.MMFNLibraryError("the MMFN HIP path needs a GPU device (got %s); there is no CPU fallback" % self.device)
        enc = module.encoder
        self.img = ResNetTrunk("img", layout, "encoder.image_encoder.features", enc.image_encoder.features)
        self.lid = ResNetTrunk("lid", layout, "encoder.lidar_encoder._model", enc.lidar_encoder._model)
        if variant == "img":
            self.map = ResNetTrunk("map", layout, "encoder.img_map_encoder.features", enc.img_map_encoder.features)
            self.vec = None
        else:
            self.map = ResNetTrunk("map", layout, "encoder.img_map_encoder.features", enc.img_map_encoder.features,
                                   with_stem=False, first_layer=2)
            self.vec = VectorNet("vec", layout, "encoder.vectornet_encoder", enc.vectornet_encoder)
        self.rad = RadarGAT("rad", layout, "encoder.radar_encoder", enc.radar_encoder) if variant == "rad" else None
        self.gpts = [GPT("gpt%d" % (i + 1), layout, "encoder.transformer%d" % (i + 1), getattr(enc, "transformer%d" % (i + 1)),
                         cfg, 100 * (i + 1)) for i in range(4)]
        for i, g in enumerate(self.gpts):
            g.frames = self.frames + ([1] if (variant == "rad" and i == 3) else [])
            if g.T != 64 * sum(g.frames):
                raise ValueError("%s.pos_emb holds %d tokens; seq_len %d / n_views %d need %d (config changed after the module was built?)"
                                 % (g.name, g.T, S, V, 64 * sum(g.frames)))
        self.head = Head(layout, cfg.pred_len)
        self.bufs = {}
        dev = self.device
        self.rng_state = torch.tensor([0x5EED, 0], dtype=torch.int64, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.norm_mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32, device=dev)
        self.norm_inv_std = torch.tensor([1.0 / s for s in IMAGENET_STD], dtype=torch.float32, device=dev)
        ops.norm_workspace(dev)
        ops.workspace(64 << 20, dev)
        # The three encoder branches (camera ResNet-34, LiDAR ResNet-18, map branch) only meet at the four
        # fusion transformers.  Between those points they run on separate HIP streams, so the many small
        # kernels of the deep stages (M = 2048..8192 rows at B = 32) overlap each other's tails and launch
        # gaps; captured into the hipGraph this becomes a fork/join DAG.
        self.side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        self.multi_stream = True
        self._recorder = None   # mmfn_amd.graphs.Recorder while a capture is running
        self.folded = {}        # ConvBN name -> (BatchNorm-folded filter, shift), see fold_batchnorm()
        # "f32" (parity path) or "bf16": bf16 MFMA operands with fp32 accumulation for the Linear / Winograd GEMMs
        self.gemm_dtype = getattr(cfg, "gemm_dtype", "f32")
        if self.act_dtype == torch.bfloat16:
            self.gemm_dtype = "f32"   # the fp32 islands of the bf16 mode (stems, VectorNet, head) are plain fp32
            self._build_shadows()
        self.ln_fold_table = None
        self.ln_fold_mode = LN_FOLD if LN_FOLD in ("eval", "1") else "0"
        if self.ln_fold_mode != "0" and self.act_dtype == torch.float32:
            self._build_ln_fold()
        self.wino_layers = {}     # ConvBN name -> (filter storage, transformed-filter buffer): filled by the first training forward
        self.wino_table = None
        self.opt_group_of = None
        self.opt_hyper = torch.zeros(16, 8, dtype=torch.float32, device=dev)
        self._hyper_host = None
        self._hyper_pinned, self._hyper_slot = None, 0
        self.n_lanes = int(os.environ.get("MMFN_BRANCH_LANES", "3"))
        self.offload_wgrad = True   # transformer weight / bias gradients on the side stream (worth 3.9 ms per step)

    # ------------------------------------------------------------------ LayerNorm folded into the Linear behind it (fp32 path)
    def ln_fold_now(self, training):
        return self.ln_fold_table is not None and (self.ln_fold_mode == "1" or not training)

    def _build_ln_fold(self):
        """Per transformer block: (Wqkv . diag(gamma1), c1, c2) and (W_mlp0 . diag(gamma2), c1, c2) for MMFN_EPI_LN_FOLD, refreshed by
        one grouped launch per forward (ops.ln_fold_weights): the weights change at every optimizer step."""
        dev, entries = self.device, []
        for gpt in self.gpts:
            C = gpt.C
            for blk in gpt.blocks:
                mk = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
                f = (mk(3 * C, C), mk(3 * C), mk(3 * C), mk(4 * C, C), mk(4 * C), mk(4 * C))
                blk["fold"] = f
                entries.append((blk["wqkv"], blk["ln1"].w, blk["ln1"].b, blk["bqkv"], f[0], f[1], f[2]))
                entries.append((blk["fc1"].w, blk["ln2"].w, blk["ln2"].b, blk["fc1"].b, f[3], f[4], f[5]))
        self.ln_fold_table = ops.make_ln_fold_table(entries, dev)

    # ------------------------------------------------------------------ bf16 weight shadows
    def _build_shadows(self):
        """bf16 mode: every GEMM weight has a bf16 shadow at the same offset of layout.params16 (forward operand) and, where a
        data gradient reads it, a transposed shadow ([in, out] / [Cin, taps, Cout]) in one extra flat buffer; both are
        re-derived from the fp32 master weights once per step (refresh_shadows: two launches)."""
        L = self.layout
        L.make_shadows()
        entries = []

        def want(src, shape_t):
            n = src.numel()
            entries.append((src, n, shape_t))

        trunks = [self.img, self.lid, self.map]
        convs = [cb for t in trunks for cb in t.convbns() if cb.w.shape[3] % 64 == 0]   # all but the 7x7 stems
        for cb in convs:
            cb.w16 = L.w16(cb.conv_name + ".weight")
            Co, KH, KW, Ci = cb.w.shape
            want(cb.w.view(Co, KH * KW, Ci), (Ci, KH, KW, Co))
        lins = []
        for gpt in self.gpts:
            for blk in gpt.blocks:
                C = gpt.C
                blk["wqkv16"] = L.packed16(blk["wqkv_name"], 3 * C, C)
                want(blk["wqkv"], (C, 3 * C))
                for k in ("proj", "fc1", "fc2"):
                    lin = blk[k]
                    lin.w16 = L.w16(lin.name)
                    want(lin.w, (lin.w.shape[1], lin.w.shape[0]))
                    lins.append(lin)
        total = sum((n + 7) // 8 * 8 for _, n, _ in entries)
        self.shadow_t = torch.zeros(total, dtype=torch.bfloat16, device=self.device)
        off, pairs, views = 0, [], []
        for src, n, shape_t in entries:
            dst = self.shadow_t[off:off + n]
            pairs.append((src, dst))
            views.append(dst.view(shape_t))
            off += (n + 7) // 8 * 8
        it = iter(views)
        for cb in convs:
            cb.w16t = next(it)
        for gpt in self.gpts:
            for blk in gpt.blocks:
                blk["wqkv16t"] = next(it)
                for k in ("proj", "fc1", "fc2"):
                    blk[k].w16t = next(it)
        self.shadow_table = ops.make_shadow_table(pairs, self.device)
        self.refresh_shadows()

    def refresh_shadows(self):
        """fp32 master weights -> bf16 operands (same offsets) + transposed copies.  Part of every bf16-mode forward: the weights
        change at every optimizer step, and a captured step must not depend on host-side bookkeeping."""
        L = self.layout
        ops.cast_to_bf16(L.params, L.params16)
        ops.shadow_transpose(*self.shadow_table)

    # ------------------------------------------------------------------ inputs
    def _bufs_for(self, B):
        b = self.bufs.get(B)
        if b is None:
            b = Buffers(self.device)
            self.bufs[B] = b
        return b

    def _ctx(self, B, training):
        cfg = self.cfg
        side = self.side[0] if (self.multi_stream and self.offload_wgrad and training) else None
        ctx = Ctx(self._bufs_for(B), training, (cfg.embd_pdrop, cfg.attn_pdrop, cfg.resid_pdrop), self.rng_state, side, engine=self)
        ctx.side2 = self.side[1] if side is not None else None
        return ctx

    def _ingest(self, ctx, inp):
        """inp: dict of device tensors -> NHWC network inputs."""
        bufs = ctx.bufs
        if "rgb_u8" in inp:
            rgb = inp["rgb_u8"]
            B = rgb.shape[0]
            img = ops.ingest_rgb_u8(rgb, bufs.get("in.img", (B, 256, 256, 3)))
        else:
            x = inp["image"]
            B = x.shape[0]
            img = ops.nchw_to_nhwc(x, bufs.get("in.img", (B, x.shape[2], x.shape[3], 3)), self.norm_mean, self.norm_inv_std)
        if "lidar_pts" in inp:
            lid = ops.lidar_splat(inp["lidar_pts"], bufs.get("in.lid", (inp["lidar_pts"].shape[0], 256, 256, 2)),
                                  flip_y=bool(inp.get("lidar_flip_y", False)))
        else:
            x = inp["lidar"]
            lid = ops.nchw_to_nhwc(x, bufs.get("in.lid", (x.shape[0], x.shape[2], x.shape[3], x.shape[1])))
        mp = None
        if self.variant == "img":
            x = inp["map"]
            mp = ops.nchw_to_nhwc(x, bufs.get("in.map", (x.shape[0], x.shape[2], x.shape[3], 3)))  # NOT normalised (model_img.py:337)
        return img, lid, mp

    # ------------------------------------------------------------------ branch concurrency
    def _branches(self, fns):
        """Run fns[0] on the current stream and fns[1:] on the side streams, fork/join with events."""
        if not self.multi_stream:
            return [f() for f in fns]
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        outs = [None] * len(fns)
        if self.n_lanes == 2:
            # two lanes: the camera ResNet-34 alone on the main stream, LiDAR ResNet-18 + map branch back to back on
            # the side stream.  Which lane count wins depends on the kernel mix: with the implicit-GEMM convolutions
            # (few, chip-filling launches) two lanes beat three by 3 %; with the Winograd path (many small streaming
            # transforms between the GEMMs) three lanes beat two by 3 %
            st = self.side[0]
            st.wait_event(fork)
            with torch.cuda.stream(st), ops.lane(1):
                for i in range(1, len(fns)):
                    outs[i] = fns[i]()
            outs[0] = fns[0]()
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)
            return outs
        for i, f in enumerate(fns[1:]):
            st = self.side[i]
            st.wait_event(fork)
            with torch.cuda.stream(st), ops.lane(i + 1):
                outs[i + 1] = f()
        outs[0] = fns[0]()
        for i in range(len(fns) - 1):
            done = torch.cuda.Event()
            done.record(self.side[i])
            main.wait_event(done)
        return outs

    # ------------------------------------------------------------------ forward / backward
    def fold_batchnorm(self):
        """(Re)compute the BatchNorm-folded filters / shifts of every convolution from the current weights and running
        statistics, for eval forwards with folded=True.  The caller owns freshness: call it again after the weights or the
        running statistics change (DrivingSession does at construction and in refresh())."""
        for trunk in (self.img, self.lid, self.map):
            for cb in trunk.convbns():
                ent = self.folded.get(cb.name)
                if ent is None:
                    ent = (torch.empty_like(cb.w), torch.empty(cb.cout, dtype=torch.float32, device=self.device),
                           torch.empty(cb.w.shape, dtype=torch.bfloat16, device=self.device) if self.act_dtype == torch.bfloat16 else None)
                    self.folded[cb.name] = ent
                ops.bn_fold(cb.w, cb.bn_w, cb.bn_b, cb.bn.running_mean, cb.bn.running_var, cb.bn.eps, ent[0], ent[1])
                if ent[2] is not None:   # bf16 mode: the folded filter's shadow (the fold itself stays fp32: w * s is rounded once)
                    ops.cast_to_bf16(ent[0], ent[2])

    @_in_precision
    def forward(self, inp, training, gt=None, folded=False):
        B = inp["target_point"].shape[0]
        ctx = self._ctx(B, training)
        if folded:
            if training or not self.folded:
                raise ValueError("folded=True is an eval-mode option and needs Engine.fold_batchnorm() first")
            ctx.folded = True
        self._last = (ctx, B)
        if ctx.bf16:
            self.refresh_shadows()
        if self.ln_fold_now(training) and ops.current_precision() == "f32":
            ops.ln_fold_weights(*self.ln_fold_table)
        img, lid, mp = self._ingest(ctx, inp)
        vel = inp["velocity"]
        trunks = [self.img, self.lid, self.map]

        if training and self.wino_layers:
            # every Winograd filter transform of the step in one launch (the layers registered themselves in an earlier
            # forward); it runs at the head of the shortest branch, the consumers (layer2 and deeper) come after the first
            # fusion transformer, where all branches have joined
            if self.wino_table is None and not torch.cuda.is_current_stream_capturing():
                # two launches: the filters of layer1 (consumed inside the first fork, by lanes that run beside the lane that
                # would transform them: they go first, on the main stream, a few microseconds) and all the others
                early = [v for k, v in self.wino_layers.items() if ".l1." in k]
                late = [v for k, v in self.wino_layers.items() if ".l1." not in k]
                self.wino_table = (ops.make_wino_group_table(early, self.device) if early else None,
                                   ops.make_wino_group_table(late, self.device) if late else None, frozenset(self.wino_layers))
            table = self.wino_table   # a layer registering during this forward resets self.wino_table: keep using this one
            ctx.wino_ready = table is not None
            if ctx.wino_ready:
                ctx.wino_names = table[2]
                if table[0] is not None:
                    ops.wino_weight_group(*table[0])
        else:
            table = None

        def map_stage1():
            if ctx.wino_ready and table[1] is not None:
                ops.wino_weight_group(*table[1])
            if self.variant == "img":
                return self.map.layer_fwd(ctx, 1, self.map.stem_fwd(ctx, mp))
            return self.vec.fwd(ctx, inp["lane"], inp["lane_num"])

        feats = self._branches([lambda: self.img.layer_fwd(ctx, 1, self.img.stem_fwd(ctx, img)),
                                lambda: self.lid.layer_fwd(ctx, 1, self.lid.stem_fwd(ctx, lid)),
                                map_stage1])
        self.taps = {"stage1": tuple(feats)}
        self.pre_add = []
        tok = None
        # token groups: the modality's first 64-token group and its frames per sample (radar: one frame after the others)
        frames, base = self.frames + [1], self.group_base + [sum(self.frames)]
        for s in range(4):
            if s > 0:
                # per branch: add the previous scale's fusion output, then the next ResNet stage
                prev, ptok = feats, tok

                def stage(m, prev=prev, ptok=ptok, s=s):
                    f = ops.upsample_add_fwd(prev[m], ptok, ctx.bufs.get("fuse%d.%d" % (s - 1, m), prev[m].shape, prev[m].dtype),
                                             base[m], frames[m])
                    return trunks[m].layer_fwd(ctx, s + 1, f)

                feats = self._branches([lambda m=m: stage(m) for m in range(3)])
            if s == 3 and self.rad is not None:  # radar joins only the deepest fusion (model_rad.py:585-593)
                feats = feats + [self.rad.fwd(ctx, inp["radar"], inp["radar_adj"])]
            tok = self.gpts[s].fwd(ctx, feats, vel)
            self.taps["gpt%d" % (s + 1)] = tok
            self.pre_add.append(feats)
        feats = [ops.upsample_add_fwd(f, tok, ctx.bufs.get("fuse3.%d" % m, f.shape, f.dtype), base[m], frames[m]) for m, f in enumerate(feats)]
        fused = ops.gap_sum_fwd(feats, ctx.bufs.get("fused", (B, 512)), frames=frames[:len(feats)])
        self.taps["fused"] = fused
        pred, loss = self.head.fwd(ctx, fused, inp["target_point"], gt)
        return pred, loss

    def backward(self, dpred=None, gscale=None, on_stage=None, on_ready=None):
        """Backward of the last training forward.  dpred None => gradient of the fused L1 loss.
        on_stage(i) is called as soon as every gradient of backward stage i (params.FlatLayout.stage_of)
        has been written, so a data-parallel wrapper can start reducing that bucket.
        on_ready((stage, group)) is the finer hook: called - ON THE STREAM THAT WROTE THEM, which inside the branch lanes is a
        side stream - as soon as the gradients of one readiness group of a stage (params.FlatLayout.group_ranges: "head",
        "gpt", "vec", "img", "lid", "map") are complete."""
        self.backward_begin(dpred, gscale, on_ready=on_ready)
        for s in range(3, -1, -1):
            self.backward_scale(s, on_ready=on_ready)
            if on_stage is not None:
                on_stage(3 - s)

    def _ready(self, hook, stage, group):
        if hook is not None and (stage, group) in self.layout.group_ranges:
            hook((stage, group))

    @_in_precision
    def backward_begin(self, dpred=None, gscale=None, on_ready=None):
        """Head backward + gradient of the global-average-pool/sum: seeds the per-branch gradients."""
        ctx, B = self._last
        bufs = ctx.bufs
        self._adj_done = False
        g_fused = self.head.bwd(ctx, dpred, gscale)
        if self.rad is None:   # (rad: the "head" group also holds the radar encoder, complete after the deepest transformer)
            self._ready(on_ready, 0, "head")
        shapes = [f.shape for f in self.pre_add[3]]
        self._G = [bufs.get("G3.%d" % m, shp, ctx.adt) for m, shp in enumerate(shapes)]
        ops.gap_sum_bwd(g_fused, self._G, frames=(self.frames + [1])[:len(self._G)])

    @_in_precision
    def backward_scale(self, s, on_ready=None):
        """Backward of fusion scale s (3 = deepest): GPT_s, then ResNet stage s+1 of the three branches
        (s = 0: layer1 + stems + VectorNet).  After it returns (enqueues), backward stage 3-s is complete."""
        ctx, B = self._last
        st = 3 - s
        names = ("img", "lid", "map")
        bufs = ctx.bufs
        trunks = [self.img, self.lid, self.map]
        G = self._G
        gpt = self.gpts[s]
        frames, base = self.frames + [1], self.group_base + [sum(self.frames)]
        gtok = bufs.get("gtok%d" % s, (B, gpt.T, gpt.C), ctx.adt)
        for m, g in enumerate(G):
            if not (self._adj_done and m < 3):   # the three branch lanes of the previous scale already spread their gradient
                ops.upsample_adj(g, gtok, base[m], frames[m])
        gin = gpt.bwd(ctx, gtok)
        self._ready(on_ready, st, "gpt")
        if s == 3 and self.rad is not None:
            dF3 = ops.pool_bcast_add(G[3], gin, bufs.get("dF3.3", G[3].shape, G[3].dtype), base[3], 1)
            self.rad.bwd(ctx, dF3)
            self._ready(on_ready, 0, "head")
        if s > 0:
            nxt = self.gpts[s - 1]
            gtok_next = bufs.get("gtok%d" % (s - 1), (B, nxt.T, nxt.C), ctx.adt)

            def stage(m):
                d = ops.pool_bcast_add(G[m], gin, bufs.get("dF%d.%d" % (s, m), G[m].shape, G[m].dtype), base[m], frames[m])
                g = trunks[m].layer_bwd(ctx, s + 1, d)
                self._ready(on_ready, st, names[m])
                # the adjoint of the next scale's upsample-add for this branch (its own 64 token rows of gtok) at the tail of
                # the lane, beside the other lanes, instead of three launches on the main stream ahead of the transformer
                ops.upsample_adj(g, gtok_next, base[m], frames[m])
                return g

            self._G = self._branches([lambda m=m: stage(m) for m in range(3)])
            self._adj_done = True
            return

        def img_tail():
            d = ops.pool_bcast_add(G[0], gin, bufs.get("dF0.0", G[0].shape, G[0].dtype), base[0], frames[0])
            self.img.stem_bwd(ctx, self.img.layer_bwd(ctx, 1, d))
            self._ready(on_ready, st, "img")

        def lid_tail():
            d = ops.pool_bcast_add(G[1], gin, bufs.get("dF0.1", G[1].shape, G[1].dtype), base[1], frames[1])
            self.lid.stem_bwd(ctx, self.lid.layer_bwd(ctx, 1, d))
            self._ready(on_ready, st, "lid")

        def map_tail():
            d = ops.pool_bcast_add(G[2], gin, bufs.get("dF0.2", G[2].shape, G[2].dtype), base[2], frames[2])
            if self.variant == "img":
                self.map.stem_bwd(ctx, self.map.layer_bwd(ctx, 1, d))
            else:
                self.vec.bwd(ctx, d)
            self._ready(on_ready, st, "map" if self.variant == "img" else "vec")

        self._branches([img_tail, lid_tail, map_tail])

    # ------------------------------------------------------------------ optimizer
    def set_param_groups(self, group_of):
        """group_of: uint8 device tensor with one optimizer-group id per 4 consecutive floats of the flat buffer
        (optim.FusedAdamW builds it from torch-style param_groups), or None = one group."""
        self.opt_group_of = group_of

    def set_hyper(self, rows):
        """rows: [(lr, beta1, beta2, eps, weight_decay, grad_scale), ...] one per optimizer group.  They live in a small
        device table read by the AdamW kernel, so a learning-rate schedule changes the table (one tiny host-to-device copy,
        outside any captured graph) instead of invalidating captured hipGraphs."""
        rows = tuple(tuple(float(x) for x in r) for r in rows)
        if rows == self._hyper_host:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("optimizer hyper-parameters changed inside a hipGraph capture; call Engine.set_hyper() before it")
        if len(rows) > 16:
            raise ValueError("at most 16 optimizer groups")
        # two pinned staging slots used alternately: the copy is asynchronous on the compute stream (a per-iteration LR schedule
        # must not block the host until the previous step has drained), and a slot is only rewritten two updates later
        if self._hyper_pinned is None:
            self._hyper_pinned = [torch.zeros(16, 8, dtype=torch.float32).pin_memory() for _ in range(2)]
            self._hyper_events = [None, None]
        slot = self._hyper_slot = (self._hyper_slot + 1) % 2
        if self._hyper_events[slot] is not None:
            self._hyper_events[slot].synchronize()   # the copy issued two updates ago has long finished
        host = self._hyper_pinned[slot]
        host.zero_()
        for i, r in enumerate(rows):
            host[i, :6] = torch.tensor(r, dtype=torch.float32)
        self.opt_hyper.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyper_events[slot] = ev
        self._hyper_host = rows

    @staticmethod
    def hyper_rows(lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0, groups=None):
        if groups is None:
            groups = [(lr, betas[0], betas[1], eps, weight_decay)]
        return [tuple(r[:5]) + (grad_scale,) for r in groups]

    def optimizer_step(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0, groups=None):
        """torch.optim.AdamW step over the trained range of the flat buffer.  `groups`: per-group
        (lr, beta1, beta2, eps, weight_decay) rows (FusedAdamW.hyper_rows()); default one group from the scalars."""
        L = self.layout
        groups = self.hyper_rows(lr, betas, eps, weight_decay, grad_scale, groups)
        self.set_hyper(groups)
        self.module.weights_changed()
        ops.step_advance(self.step_count)
        ops.adamw_groups(L.params, L.grads, L.exp_avg, L.exp_avg_sq, self.step_count, self.opt_hyper, len(groups),
                         group_of=self.opt_group_of if len(groups) > 1 else None, n=L.tail)

    def backward_and_step(self, dp=None, lr=1e-4, **adam):
        """Backward of the last training forward + AdamW (+ the gradient all-reduces of `dp`, issued from the streams that
        complete each readiness group while the backward is still running; the 1/world average is folded into AdamW)."""
        if dp is None:
            self.backward()
            self.optimizer_step(lr=lr, **adam)
            return
        dp.begin()
        self.backward(on_ready=dp.reduce)
        dp.finish()
        self.optimizer_step(lr=lr, grad_scale=1.0 / dp.world, **adam)

    def train_step(self, inp, gt, lr=1e-4, dp=None, **adam):
        """zero-grad (implicit: every gradient is overwritten) + forward + L1 + backward + AdamW
        (phase2_train_net.py:60-110).  `dp` (mmfn_amd.parallel.DataParallel) reduces the gradient
        buckets across ranks while the backward is still running.  `adam` may carry betas / eps /
        weight_decay.  Returns the device loss scalar."""
        ops.rng_advance(self.rng_state)
        _, loss = self.forward(inp, True, gt)
        self.backward_and_step(dp, lr=lr, **adam)
        return loss
