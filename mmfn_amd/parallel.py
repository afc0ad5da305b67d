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
    def __init__(self, module, dist, max_bucket_bytes=256 << 20, comm=None):
        """dist: torch.distributed (process group already initialised).  comm: optional mmfn_amd.comm.RcclComm - the
        gradient buckets then go through the C ABI (mmfn_allreduce_sum_f32) on a side HIP stream owned by this object
        instead of through torch's ProcessGroup; everything else (broadcasts, barriers) stays on `dist`."""
        self.module = module
        self.dist = dist
        self.comm = comm
        self.comm_stream = None
        self.world = dist.get_world_size()
        self.layout = module._layout
        self.buckets = []
        lim = max_bucket_bytes // 4
        for st, (b, e) in enumerate(self.layout.stage_ranges):
            e = min(e, self.layout.tail)
            chunks = []
            while b < e:
                n = min(lim, e - b)
                chunks.append((b, b + n))
                b += n
            self.buckets.append(chunks)
        self.pending = []
        # exposed-communication probe: when enabled, finish() brackets its waits with events on the compute stream; the
        # elapsed time between them is what the collectives did NOT hide behind the backward (bench.py --gpus N reports it)
        self.measure_exposed = False
        self.exposed = []

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

    def on_stage(self, stage):
        """Called by Engine.backward when every gradient of `stage` has been written (enqueued)."""
        g = self.layout.grads
        if self.comm is not None:
            # C-ABI transport: the bucket's reduction is ordered after the backward so far by an event, runs on our own
            # side stream (overlapping the rest of the backward), and finish() makes the compute stream wait for it.
            # Plain stream work, no host wait: capturable into the same hipGraph as the kernels.
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=self.layout.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            for b, e in self.buckets[stage]:
                self.comm.all_reduce_sum_(g[b:e], stream=self.comm_stream)
            self.pending.append(None)
            return
        for b, e in self.buckets[stage]:
            self.pending.append(self.dist.all_reduce(g[b:e], op=self.dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Make the compute stream wait for all outstanding reductions (before the optimizer)."""
        probe = self.measure_exposed and self.layout.device.type == "cuda"
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


class GraphedStep(object):
    """One training step as a replayable sequence of linear hipGraphs (mmfn_amd.graphs.Recorder).

    The step is cut (a) wherever the engine forks into its branch lanes - every lane is its own linear graph on its own
    stream, stitched with eager events, because replaying a graph with cross-stream edges costs the host ~5 us per kernel
    node against ~0.4 us for a linear graph (graphs.py) - and (b), under data parallelism, at the backward-stage
    boundaries, where the gradient-bucket all-reduces are issued (through torch.distributed they cannot be captured):
        RNG advance + forward + loss + head backward + backward of fusion scale 4   -> all-reduce bucket 0
        backward of scale 3 -> bucket 1;  scale 2 -> bucket 2;  scale 1 + stems -> bucket 3
        fused AdamW (after every reduction has been waited on)
    An eager step is ~2200 Python-issued launches, which makes the host the bottleneck; a replay is ~35 host calls and the
    reductions still overlap the later backward graphs."""

    def __init__(self, engine, dp, inp, gt, lr=1e-4, warm=2, lane_graphs=None, **adam):
        from .graphs import Recorder
        self.engine, self.dp = engine, dp
        eng = engine
        for _ in range(warm):  # size every buffer / scratch lane eagerly before capture
            eng.train_step(inp, gt, lr=lr, dp=dp, **adam)
        torch.cuda.synchronize()
        scale = 1.0 / (dp.world if dp is not None else 1)
        self.scale = scale
        eng.set_hyper(eng.hyper_rows(lr=lr, grad_scale=scale, **adam))  # the captured AdamW reads them from device memory
        rec = self.recorder = Recorder(eng, split_lanes=lane_graphs)   # None: MMFN_LANE_GRAPHS decides (default: forks inside the graphs)

        def body():
            from . import ops
            ops.rng_advance(eng.rng_state)
            eng.forward(inp, True, gt)
            eng.backward_begin()
            for i in range(4):
                eng.backward_scale(3 - i)
                if dp is not None and i < 3:
                    rec.cut(lambda i=i: dp.on_stage(i))
            if dp is not None:   # last bucket and the wait for all of them in one cut (nothing is launched in between)
                rec.cut(lambda: (dp.on_stage(3), dp.finish()))
            eng.optimizer_step(lr=lr, grad_scale=scale, **adam)

        rec.capture(body)
        self.loss = eng._bufs_for(inp["target_point"].shape[0]).get("head.loss", (1,))

    def set_hyper(self, lr, **adam):
        """New learning rate / Adam hyper-parameters for the following replays (no re-capture)."""
        self.engine.set_hyper(self.engine.hyper_rows(lr=lr, grad_scale=self.scale, **adam))

    def __call__(self):
        self.recorder.replay()
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
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.pred, self.loss = engine.forward(self.inp, False, self.gt)

    def __call__(self, inp, gt):
        for k, v in inp.items():
            if isinstance(v, torch.Tensor):
                self.inp[k].copy_(v, non_blocking=True)
        self.gt.copy_(gt, non_blocking=True)
        self.graph.replay()
        return self.loss
