"""Can three independent kernel chains overlap on the chip, and which replay scheme lets them?  Each lane is a chain of
[GEMM (M=8192,K=1152,N=128, ~30 us), 3 small LayerNorm launches] x 20 - the shape of a ResNet lane between two fusion
points.  Times, unprofiled: the three lanes back to back in one linear graph; one graph with the lanes forked inside;
three linear graphs on three streams stitched with events; eager launches on three streams."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmfn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, K, N, REP = int(os.environ.get("LM", 8192)), 1152, 128, 20
SMALL = int(os.environ.get("SMALL", 3))


class Lane(object):
    def __init__(self, i):
        self.i = i
        self.x = torch.randn(M, K, device=dev)
        self.w = torch.randn(N, K, device=dev) * 0.02
        self.y = torch.empty(M, N, device=dev)
        self.z = torch.empty(M, N, device=dev)
        self.g, self.b = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        self.mean, self.rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)

    def run(self):
        with ops.lane(self.i):
            for _ in range(REP):
                ops.linear_fwd(self.x, self.w, out=self.y)
                for _ in range(SMALL):
                    ops.layernorm_fwd(self.y, self.g, self.b, self.z, self.mean, self.rstd)


lanes = [Lane(i) for i in range(3)]
side = [torch.cuda.Stream(), torch.cuda.Stream()]
for l in lanes:
    l.run()
torch.cuda.synchronize()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def forked():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    for i, st in enumerate(side):
        st.wait_event(ev)
        with torch.cuda.stream(st):
            lanes[i + 1].run()
    lanes[0].run()
    for st in side:
        e = torch.cuda.Event(); e.record(st); main.wait_event(e)


g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    for l in lanes:
        l.run()
one_lane = torch.cuda.CUDAGraph()
with torch.cuda.graph(one_lane):
    lanes[0].run()
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    forked()
gl = []
for i, l in enumerate(lanes):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        l.run()
    gl.append(g)
evs = [torch.cuda.Event() for _ in range(3)]


def stitched():
    main = torch.cuda.current_stream()
    evs[0].record(main)
    for i, st in enumerate(side):
        st.wait_event(evs[0])
        with torch.cuda.stream(st):
            gl[i + 1].replay()
        evs[i + 1].record(st)
    gl[0].replay()
    main.wait_event(evs[1]); main.wait_event(evs[2])


print("lane = %d x [GEMM %dx%dx%d + %d LayerNorm launches]" % (REP, M, N, K, SMALL))
print("one lane alone, linear graph            %8.1f us" % timeit(one_lane.replay))
print("three lanes back to back, linear graph  %8.1f us" % timeit(g1.replay))
print("three lanes forked inside ONE graph     %8.1f us" % timeit(g3.replay))
print("three linear graphs + events            %8.1f us" % timeit(stitched))
print("three streams, eager launches           %8.1f us" % timeit(forked))
