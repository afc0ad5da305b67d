"""What do the data-parallel cut points cost by themselves?  The step captured with a do-nothing DataParallel stand-in (the
graphs are cut at the gradient-bucket boundaries, the hooks return at once) against the single-graph step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mmfn_amd.config import GlobalConfig
from mmfn_amd.model import MMFN
from mmfn_amd.parallel import GraphedStep

dev = torch.device("cuda:0")


class NoDP(object):
    world = 1

    def on_stage(self, i):
        pass

    def finish(self):
        pass


def run(dp, n=60):
    net = MMFN(GlobalConfig(), dev)
    inp, gt = bench.synth_inputs(32, dev, seed=0)
    step = GraphedStep(net._engine_for(), dp, inp, gt, warm=2)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, step.recorder.n_graphs


for name, dp in (("single graph", None), ("cut at the 4 bucket boundaries", NoDP()), ("single graph", None), ("cut at the 4 bucket boundaries", NoDP())):
    ms, ng = run(dp)
    print("%-34s %2d graphs  %.3f ms/step" % (name, ng, ms))
