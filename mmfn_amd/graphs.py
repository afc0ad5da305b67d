"""Replay of a captured step.

The step is captured into one hipGraph per stretch between data-parallel hooks (one graph on a single GPU, and with the C-ABI
RCCL transport whose collectives are captured too), with the branch lanes and the side work forked INSIDE it.  Two things measured
on gfx950 / ROCm 7 shape how the engine forks (profiles/r02c_graph_timeline.txt, tools/queue_overlap.py):
  * when such a graph is replayed, the first-captured child of a node stays on the node's hardware queue and later children move
    to other queues; a dependent chain that hops queues idles 10-16 us per hop.  The engine therefore forks side work once per
    transformer block (not per weight gradient) and captures the chain's next kernel BEFORE the side branch (Ctx.offload_at):
    the chain stays on one queue.  7 forks per block -> 1 late fork: 39.3 -> 37.9 ms on the same box;
  * replaying a graph with cross-stream edges costs the host ~5 us per kernel node (6-7 ms per step), a linear graph ~0.4 us.
(Round 2 also had every branch lane as a linear graph of its own, stitched with eager events: same step time once the forks were
shaped, 35.6 vs 35.2 ms; removed in round 4.)

Data-parallel hooks on the torch.distributed transport (ProcessGroup collectives cannot be captured) are cut points
(Recorder.cut): the step then is a short sequence of graphs with eager calls in between.
"""
import gc

import torch


# Retired captures.  Measured on ROCm 7.0 (tools/experiments/segv_bisect.sh): destroying a hipGraphExec at the moment
# Python's garbage collector happens to reach it - while another captured step is replaying - can leave the runtime's
# graph-launch streams dangling, and a later hipGraphLaunch of an unrelated graph dies in hip::Graph::UpdateStreams.  So a
# dropped Graph parks its CUDAGraph here, and the parked ones are destroyed at a safe point only: device idle, nothing
# capturing (drain_graveyard: before every capture and when the trainer evicts a shape).
_graveyard = []


class Graph(object):
    """torch.cuda.CUDAGraph with deferred destruction (see _graveyard)."""
    __slots__ = ("g",)

    def __init__(self):
        self.g = torch.cuda.CUDAGraph()

    def capture_begin(self, *a, **kw):
        self.g.capture_begin(*a, **kw)

    def capture_end(self):
        self.g.capture_end()

    def replay(self):
        self.g.replay()

    def __del__(self):
        g, self.g = self.g, None
        if g is not None and _graveyard is not None:   # (None: interpreter shutdown)
            _graveyard.append(g)


def drain_graveyard():
    """Destroy the retired captures.  Call with no capture in progress; waits for the device first."""
    if _graveyard:
        torch.cuda.synchronize()
        del _graveyard[:]
        torch.cuda.synchronize()


class Recorder(object):
    def __init__(self, engine):
        self.engine = engine
        self.ops = []           # ("graph", g) | ("call", fn)
        self._g = None
        self.n_graphs = 0
        self.extra_streams = []   # streams besides the engine's side streams that fork into the capture (DataParallel.comm_stream)

    # ------------------------------------------------------------------ capture
    def _begin(self):
        self._g = Graph()
        self._g.capture_begin(capture_error_mode="thread_local")   # RCCL's watchdog thread may poll events meanwhile

    def _end(self):
        if self._g is None:
            return
        self._g.capture_end()
        self.ops.append(("graph", self._g))
        self.n_graphs += 1
        self._g = None

    def capture(self, fn):
        """Run fn() once in capture mode on a private stream; returns what fn returned."""
        eng = self.engine
        if eng._recorder is not None:
            raise RuntimeError("a capture is already in progress on this engine")
        torch.cuda.synchronize()
        gc.collect()
        drain_graveyard()
        self._stream = torch.cuda.Stream(device=eng.device)
        self._stream.wait_stream(torch.cuda.current_stream())
        eng._recorder = self
        try:
            with torch.cuda.stream(self._stream):
                self._begin()
                try:
                    out = fn()
                    self._end()
                except BaseException:
                    # leave capture mode before the exception travels on: a stream (and the allocator's capture pool) left
                    # capturing makes every later synchronize() illegal, so the callers' eager fallbacks would die too
                    self._abort()
                    raise
        finally:
            eng._recorder = None
            self._g = None
        torch.cuda.current_stream().wait_stream(self._stream)
        torch.cuda.synchronize()
        return out

    def _abort(self):
        """End whatever capture is in flight on the current stream and drop the half-recorded step."""
        g, self._g = self._g, None
        if g is not None:
            # streams that were forked into this capture (branch lanes, side work, the data-parallel communication stream) and not
            # joined yet when the exception struck: join them first - ending a capture with unjoined forks fails and leaves
            # those streams in capture mode
            cur = torch.cuda.current_stream()
            for st in list(getattr(self.engine, "side", [])) + [s for s in self.extra_streams if s is not None]:
                try:
                    with torch.cuda.stream(st):
                        forked = torch.cuda.is_current_stream_capturing()
                    if forked:
                        ev = torch.cuda.Event()
                        ev.record(st)
                        cur.wait_event(ev)
                except Exception:
                    pass
            try:
                g.capture_end()
            except Exception:   # the capture may already have been invalidated by the failing call
                pass
        self.ops = []
        self.n_graphs = 0

    def cut(self, fn):
        """End the current main graph; fn() is called (eagerly, on the replaying thread) at this point of every replay."""
        self._end()
        self.ops.append(("call", fn))
        self._begin()

    # ------------------------------------------------------------------ replay
    def replay(self):
        for kind, what in self.ops:
            if kind == "graph":
                what.replay()
            else:
                what()
