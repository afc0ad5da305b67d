"""Stress check for cross-lane races: two identical models, one stepped eagerly and one through hipGraph replay (static
inputs), compared bit for bit after every step over many small-batch steps (small kernels make lane overlap likeliest)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mmfn_amd.config import GlobalConfig
from mmfn_amd.model import MMFN
from mmfn_amd.parallel import StaticBatchStep
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "1")); steps = int(os.environ.get("STEPS", "150"))
torch.manual_seed(0)
a = MMFN(GlobalConfig(), dev); b = MMFN(GlobalConfig(), dev); b.load_state_dict(a.state_dict())
a.train(); b.train()
ea, eb = a._engine_for(), b._engine_for()
batches = [bench.synth_inputs(B, dev, seed=s, lanes=16, n_lidar=4096) for s in range(8)]
inp0, gt0 = batches[0]
ea.train_step(inp0, gt0); eb.train_step(inp0, gt0)
step = StaticBatchStep(eb, None, inp0, gt0, 1e-4)
bad = 0
for i in range(steps):
    inp, gt = batches[i % len(batches)]
    la = ea.train_step(inp, gt).clone()
    lb = step(inp, gt).clone()
    torch.cuda.synchronize()
    same = torch.equal(la, lb) and torch.equal(a._layout.params, b._layout.params) and torch.equal(a._layout.buffers_flat, b._layout.buffers_flat)
    if not same:
        bad += 1
        d = (a._layout.params - b._layout.params).abs()
        print("step %d: MISMATCH loss %.9g vs %.9g, max param diff %.3e at %d" % (i, la.item(), lb.item(), d.max().item(), d.argmax().item()), flush=True)
        b.load_state_dict(a.state_dict()); b._layout.exp_avg.copy_(a._layout.exp_avg); b._layout.exp_avg_sq.copy_(a._layout.exp_avg_sq)
        eb.rng_state.copy_(ea.rng_state); eb.step_count.copy_(ea.step_count)
print("B=%d: %d steps, %d mismatches" % (B, steps, bad))
