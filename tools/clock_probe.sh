#!/bin/bash
# Sample the shader clock / power while the fp32 MFMA GEMM runs flat out (is the "157 TF/s" clock sustained?).
R=$GRAFT_REPO_ROOT
python - <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mmfn_amd import ops
x = torch.randn(16384, 4096, device="cuda"); w = torch.randn(4096, 4096, device="cuda"); y = torch.empty(16384, 4096, device="cuda")
for _ in range(5): ops.linear_fwd(x, w, out=y)
torch.cuda.synchronize(); t0 = time.time(); n = 0
while time.time() - t0 < 12:
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(50): ops.linear_fwd(x, w, out=y)
    e1.record(); torch.cuda.synchronize(); n += 1
    if n % 8 == 0: print("gemm 16384x4096x4096: %.1f TF/s" % (50 * 2 * 16384 * 4096 * 4096 / (e0.elapsed_time(e1) * 1e-3) / 1e12), flush=True)
PY
sleep 6
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power|fclk|mclk" | head -6; sleep 1; done
wait
