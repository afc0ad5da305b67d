"""Unprofiled time of every linear graph of the lane-graph step, replayed on its own (synchronised), next to the whole
step: which stretches of the step are single-lane (fusion transformers), what the lanes cost alone and overlapped."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from mmfn_amd.config import GlobalConfig  # noqa: E402
from mmfn_amd.model import MMFN  # noqa: E402
from mmfn_amd.parallel import GraphedStep  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "32"))
net = MMFN(GlobalConfig(gemm_dtype=os.environ.get("GEMM_DTYPE", "f32"), act_dtype=os.environ.get("ACT_DTYPE", "f32")), dev)
inp, gt = bench.synth_inputs(B, dev, seed=0)
step = GraphedStep(net._engine_for(), None, inp, gt, warm=2, lane_graphs=True)
rec = step.recorder


def t_graph(g, n=10):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
whole = (time.perf_counter() - t0) / 20 * 1e3
tot_main = tot_lane_sum = tot_lane_max = 0.0
i = 0
ops_ = rec.ops
out = []
while i < len(ops_):
    op = ops_[i]
    if op[0] == "graph":
        t = t_graph(op[1]); tot_main += t
        out.append("main graph            %7.3f ms" % t)
        i += 1
    elif op[0] == "lanes":
        ts = [t_graph(g) for _, g, _ in op[2]]
        t0_ = t_graph(ops_[i + 1][1])      # the main-lane graph follows
        allt = [t0_] + ts
        tot_lane_sum += sum(allt); tot_lane_max += max(allt)
        out.append("lanes (main, side...)  " + "  ".join("%7.3f" % x for x in allt) + "   sum %7.3f  max %7.3f" % (sum(allt), max(allt)))
        i += 2
    else:
        i += 1
print("\n".join(out))
print("whole step %.2f ms;  main-only graphs %.2f ms;  lanes: sum %.2f ms, max-per-fork %.2f ms" % (whole, tot_main, tot_lane_sum, tot_lane_max))
print("serial estimate %.2f ms, perfect-lane-overlap estimate %.2f ms" % (tot_main + tot_lane_sum, tot_main + tot_lane_max))
