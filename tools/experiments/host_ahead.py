"""How far ahead of the GPU does the host get while replaying the captured step?  Prints the host time of consecutive
replay calls (no synchronisation in between) and the synchronised step time; plus the launch cost of a linear graph of
N tiny kernels and of the same kernels on three streams."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from mmfn_amd import ops  # noqa: E402
from mmfn_amd.config import GlobalConfig  # noqa: E402
from mmfn_amd.model import MMFN  # noqa: E402
from mmfn_amd.parallel import GraphedStep  # noqa: E402

dev = torch.device("cuda:0")


def tiny_graph(n, streams):
    cnt = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(max(1, streams))]
    side = [torch.cuda.Stream() for _ in range(streams - 1)] if streams > 1 else []
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main)
        for i, st in enumerate(side):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                for _ in range(n // streams):
                    ops.step_advance(cnt[i + 1])
        for _ in range(n // max(1, streams)):
            ops.step_advance(cnt[0])
        for st in side:
            e = torch.cuda.Event(); e.record(st); main.wait_event(e)
    return g


for streams in (1, 3):
    g = tiny_graph(999, streams)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("999 tiny kernels, %d stream(s): replay call %.3f ms, until done %.3f ms (%.2f us / kernel)" % (streams, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e6 / 999))

B = 32
net = MMFN(GlobalConfig(), dev)
inp, gt = bench.synth_inputs(B, dev, seed=0)
eng = net._engine_for()
step = GraphedStep(eng, None, inp, gt, warm=2)
print("graphs in the step:", step.recorder.n_graphs)
for _ in range(3):
    step()
torch.cuda.synchronize()
ts = [time.perf_counter()]
for _ in range(8):
    step()
    ts.append(time.perf_counter())
torch.cuda.synchronize()
te = time.perf_counter()
print("host time of consecutive replay calls (ms):", " ".join("%.2f" % ((b - a) * 1e3) for a, b in zip(ts, ts[1:])))
print("all 8 done after %.2f ms (%.2f ms / step)" % ((te - ts[0]) * 1e3, (te - ts[0]) * 1e3 / 8))
