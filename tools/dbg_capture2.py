import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
from mmfn_amd.config import GlobalConfig
from mmfn_amd.model import MMFN
import bench
dev = torch.device("cuda:0")
net = MMFN(GlobalConfig(), dev); net.train()
inp, gt = bench.synth_inputs(4, dev, 1)
eng = net._engine_for()
eng.train_step(inp, gt); eng.train_step(inp, gt); torch.cuda.synchronize()
def try_cap(name, fn):
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize(); print(name, "OK")
    except Exception as e:
        print(name, "FAILED:", str(e).split("\n")[0])
        torch.cuda.synchronize()
orig = eng._branches
calls = []
def counting(fns):
    calls.append(len(fns)); return orig(fns)
eng._branches = counting
try_cap("forward", lambda: eng.forward(inp, True, gt))
print(calls); calls.clear()
def fb():
    eng.forward(inp, True, gt); eng.backward()
try_cap("fwd+bwd", fb)
print(calls)
