"""Replay of a captured step, and the optional "lane graph" scheme.

Default (MMFN_LANE_GRAPHS=0): the step is captured into one hipGraph per stretch between data-parallel hooks (one graph on a
single GPU), with the branch lanes and the side work forked INSIDE it.  Two things measured on gfx950 / ROCm 7 shape how the
engine forks (profiles/r02c_graph_timeline.txt, tools/queue_overlap.py):
  * when such a graph is replayed, the first-captured child of a node stays on the node's hardware queue and later children move
    to other queues; a dependent chain that hops queues idles 10-16 us per hop.  The engine therefore forks side work once per
    transformer block (not per weight gradient) and captures the chain's next kernel BEFORE the side branch (Ctx.offload_at):
    the chain stays on one queue.  7 forks per block -> 1 late fork: 39.3 -> 37.9 ms on the same box;
  * replaying a graph with cross-stream edges costs the host ~5 us per kernel node (6-7 ms per step), a linear graph ~0.4 us.

Lane graphs (MMFN_LANE_GRAPHS=1): every branch lane and every piece of side work is its own LINEAR hipGraph on its own stream,
stitched with eager HIP events:

    main graph | fork event | lane graph on side stream 0 | lane graph on side stream 1 | main-lane graph | join | ...

Exact dependencies (the side work of a transformer block starts the moment the block's chain is through) and 1.5-2.8 ms of
host time per step instead of 6-7, but 40-90 us of idle time wherever the main stream passes from one graph to the next:
35.6 vs 35.2 ms per step on one GPU (tools/ab_bench.sh), so it is the option, not the default; for lanes of small kernels
(tools/experiments/lane_overlap.py: 3 x 20 x [64-tile GEMM + 3 LayerNorms]) it wins, 845 vs 1092 us.

Engine._branches() / Ctx.offload() / Ctx.rejoin() call Recorder.branches() / side() / join() while a lane-graph Recorder is
attached.  Data-parallel hooks (gradient-bucket all-reduces through torch.distributed, which cannot be captured) are cut
points in both schemes.
"""
import gc
import os

import torch


# Retired captures.  Measured on ROCm 7.0 (tools/experiments/segv_bisect.sh): destroying a hipGraphExec at the moment
# Python's garbage collector happens to reach it - while another captured step is replaying - can leave the runtime's
# graph-launch streams dangling, and a later hipGraphLaunch of an unrelated graph dies in hip::Graph::UpdateStreams.  So a
# dropped Graph parks its CUDAGraph here, and the parked ones are destroyed at a safe point only: device idle, nothing
# capturing (drain_graveyard: before every capture and when the trainer evicts a shape).
_graveyard = []
_KEEP_FOREVER = os.environ.get("MMFN_KEEP_GRAPHS") == "1"   # experiment: never destroy a captured graph


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
    if _graveyard and not _KEEP_FOREVER:
        torch.cuda.synchronize()
        del _graveyard[:]
        torch.cuda.synchronize()


class Recorder(object):
    def __init__(self, engine, split_lanes=None):
        self.engine = engine
        # MMFN_LANE_GRAPHS=1: every branch lane / piece of side work its own linear graph (module docstring); default: forks inside
        # the main graphs
        self.split_lanes = (os.environ.get("MMFN_LANE_GRAPHS", "0") == "1") if split_lanes is None else bool(split_lanes)
        self.ops = []           # ("graph", g) | ("lanes", fork, [(stream, g, done), ...]) | ("join", [events]) | ("side", ev, stream, g) | ("wait", ev, stream) | ("call", fn)
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
            raise RuntimeError("a lane-graph capture is already in progress on this engine")
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

    @staticmethod
    def _captured(graph, fn):
        """fn() captured into `graph` on the current stream; the capture is ended even when fn raises."""
        graph.capture_begin(capture_error_mode="thread_local")
        try:
            out = fn()
        except BaseException:
            try:
                graph.capture_end()
            except Exception:
                pass
            raise
        graph.capture_end()
        return out

    def cut(self, fn):
        """End the current main graph; fn() is called (eagerly, on the replaying thread) at this point of every replay."""
        self._end()
        self.ops.append(("call", fn))
        self._begin()

    def branches(self, groups):
        """groups[0]: callables for the main stream; groups[i>0] = (side stream, [callables]).  Returns the callables'
        results in order.  Each group becomes one linear graph; the main stream waits for the side lanes afterwards."""
        from . import ops
        self._end()
        fork = torch.cuda.Event()
        lanes, outs = [], []
        side_outs = []
        for lane_id, (st, fns) in enumerate(groups[1:], start=1):
            g = Graph()
            with torch.cuda.stream(st), ops.lane(lane_id):
                side_outs.append(self._captured(g, lambda fns=fns: [f() for f in fns]))
            lanes.append((st, g, torch.cuda.Event()))
            self.n_graphs += 1
        self.ops.append(("lanes", fork, lanes))
        self._begin()
        outs = [f() for f in groups[0]]
        self._end()
        self.ops.append(("join", [d for _, _, d in lanes]))
        self._begin()
        for so in side_outs:
            outs.extend(so)
        return outs

    def side(self, stream, fn, lane_id=1):
        """fn's launches (work that only feeds the optimizer) as a linear graph of their own on `stream`, ordered after
        everything the main stream has captured so far and NOT joined back here: the main stream continues at once, a later
        join(stream) waits for it.  Inside one captured graph such a fork is replayed with coarse dependencies - measured: the
        side work of all eight blocks of a transformer started only when the chain was nearly through, and every fork moved
        the chain to another hardware queue (10-16 us idle per hop); an eager event between two linear graphs is exact."""
        from . import ops
        self._end()
        g = Graph()
        with torch.cuda.stream(stream), ops.lane(lane_id):
            self._captured(g, fn)
        self.n_graphs += 1
        self.ops.append(("side", torch.cuda.Event(), stream, g))
        self._begin()

    def join(self, stream):
        """The main stream waits for everything replayed on `stream` so far."""
        self._end()
        self.ops.append(("wait", torch.cuda.Event(), stream))
        self._begin()

    # ------------------------------------------------------------------ replay
    def replay(self):
        main = torch.cuda.current_stream()
        for op in self.ops:
            kind = op[0]
            if kind == "graph":
                op[1].replay()
            elif kind == "lanes":
                fork = op[1]
                fork.record(main)
                for st, g, done in op[2]:
                    st.wait_event(fork)
                    with torch.cuda.stream(st):
                        g.replay()
                    done.record(st)
            elif kind == "join":
                for d in op[1]:
                    main.wait_event(d)
            elif kind == "side":
                _, ev, st, g = op
                ev.record(main)
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    g.replay()
            elif kind == "wait":
                op[1].record(op[2])
                main.wait_event(op[1])
            else:
                op[1]()
