"""Lane graphs: a step replayed as a sequence of LINEAR hipGraphs, one per branch lane and per stretch of the main
stream, stitched with eager HIP events.

    main graph | fork event | lane graph on side stream 0 | lane graph on side stream 1 | main-lane graph | join | ...

Why not one hipGraph with the branch streams forked inside it (round 1): measured on gfx950 / ROCm 7
(tools/experiments/host_ahead.py, lane_overlap.py), replaying a graph that contains cross-stream edges costs the host
~5 us per kernel node (the runtime walks the DAG and wires signals at launch), a linear graph ~0.4 us (pre-built packets
copied into the queue).  For the 2227-kernel training step that is 6.0-7.7 ms of host time per replay against 3.1-4.4 ms
for the 33 linear graphs - it matters once eight ranks share the node's cores - and for lanes of small kernels the forked
graph is host-bound outright (three lanes of 20 x [64-tile GEMM + 3 LayerNorms]: 1092 us forked, 845 us stitched).  On the
single-GPU training step both schemes replay in the same 37.4 ms: there the lanes overlap as far as the chip lets them
(tools/experiments/segment_times.py: lanes alone 20.0 ms, overlapped 15.7; a chip-filling GEMM of one lane leaves the
other lanes' kernels waiting for CUs either way).

Engine._branches() calls Recorder.branches() while a Recorder is attached; everything else (the fusion transformers, whose
weight-gradient GEMMs fork to the side stream and rejoin inside one graph) is captured into the current main graph.
Data-parallel hooks (gradient-bucket all-reduces through torch.distributed, which cannot be captured) are cut points too.
MMFN_LANE_GRAPHS=0 keeps the lanes as forks inside the main graphs (A/B switch).
"""
import gc
import os

import torch


class Recorder(object):
    def __init__(self, engine, split_lanes=None):
        self.engine = engine
        # MMFN_LANE_GRAPHS=0: keep the branch lanes as forks inside one graph (the round-1 scheme, for A/B measurements)
        self.split_lanes = (os.environ.get("MMFN_LANE_GRAPHS", "1") == "1") if split_lanes is None else bool(split_lanes)
        self.ops = []           # ("graph", g) | ("lanes", fork_event, [(stream, g, done_event), ...]) | ("join", [events]) | ("call", fn)
        self._g = None
        self.n_graphs = 0

    # ------------------------------------------------------------------ capture
    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
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
        self._stream = torch.cuda.Stream(device=eng.device)
        self._stream.wait_stream(torch.cuda.current_stream())
        eng._recorder = self
        try:
            with torch.cuda.stream(self._stream):
                self._begin()
                out = fn()
                self._end()
        finally:
            eng._recorder = None
            self._g = None
        torch.cuda.current_stream().wait_stream(self._stream)
        torch.cuda.synchronize()
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
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st), ops.lane(lane_id):
                g.capture_begin(capture_error_mode="thread_local")
                side_outs.append([f() for f in fns])
                g.capture_end()
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
            else:
                op[1]()
