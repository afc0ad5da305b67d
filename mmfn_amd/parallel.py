"""Data parallelism over the GPUs of one node: one process per GPU, RCCL over xGMI.

Replaces torch DDP as used by the reference (run_steps/phase2_train_net.py:227,265-269; the
reference's wiring is broken for ranks != 0).  Design for MI355X:
  * gradients already live in one flat fp32 buffer ordered by backward stage
    (params.FlatLayout.stage_of), so there is nothing to copy into buckets: each stage is one
    contiguous range, all-reduced in place as soon as the engine reports the stage complete;
  * the first bucket (fusion scale 4: ~60 % of all parameters) is ready after ~25 % of the
    backward, so its reduction overlaps the remaining ~75 %; RCCL runs on its own stream;
  * the never-trained tail of the buffer (vec/rad: raster-map stem/layer1) is statically excluded
    instead of DDP's find_unused_parameters=True;
  * the 1/world averaging is folded into the fused AdamW (grad_scale), no extra pass;
  * BatchNorm statistics stay per-GPU exactly like the reference (no SyncBN); parameters and
    buffers are broadcast from rank 0 once at start.
"""
import torch


class DataParallel(object):
    def __init__(self, module, dist, max_bucket_bytes=64 << 20, comm=None, grad_dtype=None):
        """grad_dtype: "f32" (default in every mode: what the reference's DDP exchanges, also under torch.autocast -
        phase2_train_net.py:227,265-269) or "bf16" = an explicit opt-in (bench.py --grad-dtype bf16): the buckets cross the
        links as bf16, 210 MB instead of 419 MB per step: each bucket is cast into a bf16 staging range, summed there (rounding
        at every hop of the ring), and the sum cast back into the fp32 gradient buffer - on the communication stream, behind the
        event that marks the bucket complete - so AdamW, the master weights and the moments stay fp32.  Its error against
        the fp32 exchange is bounded in tests/test_parallel_cpu.py.

        dist: torch.distributed (process group already initialised).  comm: optional mmfn_amd.comm.RcclComm - the
        gradient buckets then go through the C ABI (mmfn_allreduce_sum_f32) on a side HIP stream owned by this object
        instead of through torch's ProcessGroup; everything else (broadcasts, barriers) stays on `dist`.

        Buckets: one per (backward stage, readiness group) of the flat gradient buffer (params.FlatLayout.group_ranges: the
        fusion transformer of a scale, then each trunk's ResNet stage, VectorNet, ... - 17 ranges for the vec variant), cut
        into chunks of at most max_bucket_bytes; the engine reports each group complete on the stream that wrote it
        (Engine.backward(on_ready=dp.reduce)), so e.g. VectorNet's 66 MB leave while the camera trunk's layer1 is still
        back-propagating and what is exposed after the last backward kernel is < 1 MB."""
        self.module = module
        self.dist = dist
        self.comm = comm
        self.comm_stream = None
        self.world = dist.get_world_size()
        self.layout = module._layout
        if grad_dtype is None:
            grad_dtype = "f32"
        if grad_dtype not in ("f32", "bf16"):
            raise ValueError("grad_dtype must be f32 or bf16, got %r" % (grad_dtype,))
        self.grad_dtype = grad_dtype
        self.g16 = None   # bf16 staging for the buckets (same offsets as the flat gradient buffer), allocated on first use
        if comm is not None and self.layout.device is not None and self.layout.device.type == "cuda":
            self.comm_stream = torch.cuda.Stream(device=self.layout.device)
        lim = max(1, max_bucket_bytes // 4)
        L = self.layout

        def chunks_of(b, e):
            e = min(e, L.tail)
            out = []
            while b < e:
                n = min(lim, e - b)
                out.append((b, b + n))
                b += n
            return out

        self.groups = {}         # (stage, group name) -> [(begin, end), ...]
        self.group_order = []    # storage order == readiness order inside a stage
        for key, (b, e) in sorted(L.group_ranges.items(), key=lambda kv: kv[1][0]):
            self.groups[key] = chunks_of(b, e)
            self.group_order.append(key)
        # per-stage view (on_stage(), bench.py's byte count): all chunks of the stage's groups
        self.buckets = [[c for key in self.group_order if key[0] == st for c in self.groups[key]] for st in range(4)]
        self.pending = []
        self.reduced = set()     # groups reduced since the last finish(): a group is reduced at most once per step
        # exposed-communication probe: when enabled, finish() brackets its waits with events on the compute stream; the
        # elapsed time between them is what the collectives did NOT hide behind the backward (bench.py --gpus N reports it)
        self.measure_exposed = False
        self.exposed = []

    def n_buckets(self):
        return sum(len(c) for c in self.groups.values())

    def broadcast_parameters(self, src=0):
        L = self.layout
        self.dist.broadcast(L.params, src)
        self.dist.broadcast(L.buffers_flat, src)
        self.dist.broadcast(L.counters_flat, src)
        # optimizer moments + step counter too, so a run resumed on rank 0 continues identically everywhere
        self.dist.broadcast(L.exp_avg, src)
        self.dist.broadcast(L.exp_avg_sq, src)
        if L.device.type == "cuda":  # (the CPU unit tests exercise the bucket logic without an engine)
            eng = self.module._engine_for()
            self.dist.broadcast(eng.step_count, src)
            self.dist.broadcast(eng.rng_state, src)
            eng.rng_state[0] += self.dist.get_rank()  # per-rank dropout streams

    def _staging(self):
        if self.g16 is None:
            L = self.layout
            self.g16 = torch.empty((L.tail + 7) // 8 * 8, dtype=torch.bfloat16, device=L.grads.device)
        return self.g16

    def _narrow(self, b, e):
        """fp32 bucket -> its bf16 staging range (on the current stream)."""
        g, s = self.layout.grads, self._staging()
        if g.is_cuda:
            from . import ops
            ops.cast_to_bf16(g[b:e], s[b:e])
        else:
            s[b:e].copy_(g[b:e])     # (CPU: the unit tests of the bucket logic)
        return s[b:e]

    def _widen(self, b, e):
        g, s = self.layout.grads, self._staging()
        if g.is_cuda:
            from . import ops
            ops.cast_to_f32(s[b:e], g[b:e])
        else:
            g[b:e].copy_(s[b:e])

    def bytes_per_step(self):
        return (2 if self.grad_dtype == "bf16" else 4) * sum(e - b for chunks in self.buckets for b, e in chunks)

    def _reduce_chunks(self, chunks):
        g = self.layout.grads
        if not chunks:
            return
        half = self.grad_dtype == "bf16"
        if half and any((b % 4) or (e % 4) for b, e in chunks):
            raise ValueError("bf16 gradient buckets need 4-element aligned ranges")
        if self.comm is not None:
            # C-ABI transport: the reduction is ordered after the work enqueued so far on the CURRENT stream (the stream that
            # wrote these gradients - inside the engine's branch lanes a side stream) by an event, runs on our own
            # communication stream (overlapping the rest of the backward), and finish() makes the compute stream wait for it.
            # Plain stream work, no host wait: capturable into the same hipGraph as the kernels.
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=self.layout.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            if half:
                with torch.cuda.stream(self.comm_stream):
                    for b, e in chunks:
                        self.comm.all_reduce_sum_(self._narrow(b, e), stream=self.comm_stream)
                        self._widen(b, e)
            else:
                for b, e in chunks:
                    self.comm.all_reduce_sum_(g[b:e], stream=self.comm_stream)
            self.pending.append(None)
            return
        for b, e in chunks:
            if half:
                # (the cast runs on the stream that wrote the bucket; ProcessGroupNCCL orders its collective after that stream)
                self.pending.append((self.dist.all_reduce(self._narrow(b, e), op=self.dist.ReduceOp.SUM, async_op=True), b, e))
            else:
                self.pending.append(self.dist.all_reduce(g[b:e], op=self.dist.ReduceOp.SUM, async_op=True))

    def begin(self):
        """Start of a step's backward (Engine.backward_and_step, GraphedStep.__call__).  A step that raised between its first
        reduce() and finish() - out of memory, a failed capture that the caller retries eagerly - must not leak into the next
        one: groups still marked as reduced would be skipped (training on un-reduced gradients, or ranks issuing different
        collective sequences and hanging).  Outstanding ProcessGroup work of such a step is completed first."""
        if self.pending and self.comm is None:
            for w in self.pending:
                if w is not None:
                    (w[0] if isinstance(w, tuple) else w).wait()
        self.pending = []
        self.reduced = set()

    def reduce(self, key):
        """Engine hook (Engine.backward(on_ready=...)): every gradient of readiness group `key` = (stage, name) has been
        enqueued on the current stream.  Every rank calls this in the same program order, so the collectives match up."""
        if key in self.reduced or key not in self.groups:
            return
        self.reduced.add(key)
        self._reduce_chunks(self.groups[key])

    def on_stage(self, stage):
        """Coarse hook: every gradient of backward `stage` has been written (enqueued): reduces whatever groups of the
        stage reduce() has not been called for yet."""
        for key in self.group_order:
            if key[0] == stage:
                self.reduce(key)

    def finish(self):
        """Make the compute stream wait for all outstanding reductions (before the optimizer)."""
        for key in self.group_order:   # nothing may be left behind (a variant whose engine does not report some group)
            self.reduce(key)
        self.reduced = set()
        probe = self.measure_exposed and self.layout.device.type == "cuda" and not torch.cuda.is_current_stream_capturing()
        if probe:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.comm is not None:
            if self.pending:
                done = torch.cuda.Event()
                done.record(self.comm_stream)
                torch.cuda.current_stream().wait_event(done)
        else:
            for w in self.pending:
                if isinstance(w, tuple):     # a bf16 bucket: the sum goes back into the fp32 gradient buffer
                    w[0].wait()
                    self._widen(w[1], w[2])
                else:
                    w.wait()
        self.pending = []
        if probe:
            e1.record()
            self.exposed.append((e0, e1))

    def exposed_ms(self):
        """Mean milliseconds per step the compute stream sat waiting for gradient reductions (needs measure_exposed)."""
        if not self.exposed:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.exposed]
        self.exposed = []
        return sum(ms) / len(ms)


def connect(module, dist, transport=None, max_bucket_bytes=64 << 20, grad_dtype=None, transport_opts=None):
    """DataParallel on the best transport that comes up on EVERY rank - what trainer.fit / bench.py --gpus N use.

    transport (or the MMFN_DP_TRANSPORT environment variable): "torch" (default): torch.distributed (backend "nccl" = RCCL), the step
    cut at the bucket boundaries - the conservative choice for a first run on real multi-GPU hardware: RCCL has never seen more than
    one rank under the C-ABI transport, and a collective captured into a hipGraph that deadlocks on replay cannot be recovered from
    inside the process; "auto": the C-ABI RCCL communicator (mmfn_amd.comm) when the library loads, the communicator initialises and a
    self-test all-reduce returns the right sum on every rank - the step is then ONE hipGraph with the collectives captured inside
    (GraphedStep) - otherwise torch.distributed; "capi": the C ABI or an error.  Returns (DataParallel, note or None).
    transport_opts: keyword arguments for comm.open_transport (timeout_s; make_id / make_comm stand-ins in the CPU tests of the
    fallback protocol, which also lift the "device must be a GPU" condition)."""
    import os
    from . import comm as C
    want = transport or os.environ.get("MMFN_DP_TRANSPORT", "torch")
    if want not in ("auto", "capi", "torch"):
        raise ValueError("transport must be auto / capi / torch, got %r" % (want,))
    dev = module._layout.device
    handle, note = None, None
    opts = dict(transport_opts or {})
    if want != "torch" and dist.get_world_size() >= 1 and (dev.type == "cuda" or "make_comm" in opts):
        if want == "auto" and os.environ.get("MMFN_BENCH_SINGLE_DEVICE"):
            note = "single-device CI run: RCCL refuses two ranks on one GPU, torch.distributed (gloo) instead"
        else:
            handle, note = C.open_transport(dist.get_rank(), dist.get_world_size(), dist, dev, required=(want == "capi"), **opts)
    return DataParallel(module, dist, max_bucket_bytes=max_bucket_bytes, comm=handle, grad_dtype=grad_dtype), note


class GraphedStep(object):
    """One training step as a replayable hipGraph (or a short sequence of them; mmfn_amd.graphs.Recorder).

    Single GPU: one graph, the branch lanes and the side work forked inside it.  Data parallel, by transport:
      * C ABI (DataParallel(comm=RcclComm)): the bucket all-reduces are plain work on a HIP stream of ours
        (mmfn_allreduce_sum_f32), so they are CAPTURED - the whole data-parallel step, collectives included, is ONE graph: a
        replay is one host call, which is what eight ranks sharing one host's cores need;
      * torch.distributed: a ProcessGroup collective cannot be captured, so the step is cut at the four backward-stage
        boundaries and the buckets of the stage (one per readiness group, DataParallel.groups) are issued between the replays:
            RNG advance + forward + loss + head backward + backward of fusion scale 4   -> buckets of stage 0
            backward of scale 3 -> stage 1;  scale 2 -> stage 2;  scale 1 + stems + VectorNet -> stage 3, wait for all
            fused AdamW
    An eager step is ~2200 Python-issued launches, which makes the host the bottleneck."""

    def __init__(self, engine, dp, inp, gt, lr=1e-4, warm=2, single_graph=None, **adam):
        from .graphs import Recorder
        self.engine, self.dp = engine, dp
        eng = engine
        for _ in range(warm):  # size every buffer / scratch lane eagerly before capture
            eng.train_step(inp, gt, lr=lr, dp=dp, **adam)
        torch.cuda.synchronize()
        scale = 1.0 / (dp.world if dp is not None else 1)
        self.scale = scale
        eng.set_hyper(eng.hyper_rows(lr=lr, grad_scale=scale, **adam))  # the captured AdamW reads them from device memory
        rec = self.recorder = Recorder(eng)
        if dp is not None:
            rec.extra_streams.append(dp.comm_stream)
            if dp.comm is not None:
                dp.comm.retain(rec)   # the communicator must outlive the capture (RcclComm.destroy)
        if single_graph is None:
            single_graph = dp is not None and dp.comm is not None
        if single_graph and (dp is None or dp.comm is None):
            raise ValueError("a single-graph data-parallel step needs the C-ABI transport (DataParallel(comm=...))")
        self.single_graph = bool(single_graph) or dp is None

        def body():
            from . import ops
            ops.rng_advance(eng.rng_state)
            eng.forward(inp, True, gt)
            if dp is None or single_graph:
                # single GPU, or collectives captured on the communication stream (forked from the stream that wrote each
                # bucket, joined by finish()): no cut, one replay call per step
                eng.backward_and_step(dp, lr=lr, **adam)
                return
            else:
                tags = []
                eng.backward_begin(on_ready=tags.append)
                for i in range(4):
                    eng.backward_scale(3 - i, on_ready=tags.append)
                    mine, tags[:] = list(tags), []
                    if i < 3:
                        rec.cut(lambda mine=mine, i=i: ([dp.reduce(t) for t in mine], dp.on_stage(i)))
                    else:   # last buckets and the wait for all of them in one cut (nothing is launched in between)
                        rec.cut(lambda mine=mine: ([dp.reduce(t) for t in mine], dp.finish()))
            eng.optimizer_step(lr=lr, grad_scale=scale, **adam)

        rec.capture(body)
        self.loss = eng._bufs_for(inp["target_point"].shape[0]).get("head.loss", (1,))

    def set_hyper(self, lr, **adam):
        """New learning rate / Adam hyper-parameters for the following replays (no re-capture)."""
        self.engine.set_hyper(self.engine.hyper_rows(lr=lr, grad_scale=self.scale, **adam))

    def __call__(self):
        if self.dp is not None and not self.single_graph:
            self.dp.begin()   # (single graph: the bucket bookkeeping only ran at capture time)
        self.recorder.replay()
        self.engine.module.weights_changed()   # the replayed AdamW does not pass through Engine.optimizer_step
        return self.loss


class StaticBatchStep(object):
    """A replayable training step for a stream of different batches of one shape: the step is captured once over static
    copies of the inputs (GraphedStep: linear hipGraphs per lane and per gradient bucket) and every call
    copies the new batch into them first.  What a real training loop needs to run at the replay rate of bench.py instead
    of being bound by ~2500 Python-issued launches per step.  The engine's buffers for this shape must already exist
    (run one eager step of the shape first).  lr / betas / eps / weight decay (per optimizer group) are NOT baked into the
    capture: the AdamW kernel reads them from a device table that __call__ refreshes when they change."""

    def __init__(self, engine, dp, inp, gt, lr, **adam):
        self.engine = engine
        self.inp = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
        self.gt = gt.clone()
        self.scale = 1.0 / (dp.world if dp is not None else 1)
        engine.set_hyper(engine.hyper_rows(lr=lr, grad_scale=self.scale, **adam))
        torch.cuda.synchronize()
        self.seg = GraphedStep(engine, dp, self.inp, self.gt, lr=lr, warm=0, **adam)
        self.loss = self.seg.loss
        self.run = self.seg

    @staticmethod
    def signature(inp, gt):
        """Shapes only: hyper-parameters live in device memory and are not part of a capture's identity."""
        items = tuple(sorted((k, tuple(v.shape) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()))
        return items, tuple(gt.shape)

    def __call__(self, inp, gt, lr=None, **adam):
        if lr is not None:
            self.engine.set_hyper(self.engine.hyper_rows(lr=lr, grad_scale=self.scale, **adam))
        for k, v in inp.items():
            if isinstance(v, torch.Tensor):
                self.inp[k].copy_(v, non_blocking=True)
        self.gt.copy_(gt, non_blocking=True)
        self.run()
        return self.loss


class StaticEvalStep(object):
    """Validation counterpart of StaticBatchStep: the eval-mode forward + L1 loss (phase2_train_net.py:124-177) captured once
    per input shape over static copies of the inputs; every later batch of that shape copies its inputs and replays one graph
    instead of ~700 Python-issued launches.  The weights are read from the flat buffer at replay time, so the same capture
    serves every validation pass of a run."""

    def __init__(self, engine, inp, gt):
        self.engine = engine
        self.inp = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
        self.gt = gt.clone()
        from . import graphs
        torch.cuda.synchronize()
        graphs.drain_graveyard()
        self.graph = graphs.Graph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.pred, self.loss = engine.forward(self.inp, False, self.gt)

    def __call__(self, inp, gt):
        for k, v in inp.items():
            if isinstance(v, torch.Tensor):
                self.inp[k].copy_(v, non_blocking=True)
        self.gt.copy_(gt, non_blocking=True)
        self.graph.replay()
        return self.loss
