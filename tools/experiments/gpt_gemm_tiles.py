"""Which tile serves the transformers' GEMMs (M = 6144 rows at B = 32)?  Every Linear of transformer 3 / 4 in its three forms
(forward NK, data gradient KN, weight gradient TN) under tile = 0 (tuning table / heuristic), 1 = 128x128, 2 = 64x64, 3 = 128x64, 4 = 64x128,
isolated back-to-back launches (warm L2: an upper bound on what the step sees)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmfn_amd import ops
dev = "cuda:0"
M = 6144
iters = 30


def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for C in (512, 256):
    for name, N, K in (("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); dy = torch.randn(M, N, device=dev)
        y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
        fl = 2.0 * M * N * K
        for form, fn in (("fwd", lambda t: ops.linear_fwd(x, w, out=y, tile=t)), ("dx ", lambda t: ops.linear_dx(dy, w, out=dx, tile=t)),
                         ("dw ", lambda t: ops.linear_dw(dy, x, out=dw, tile=t))):
            row = []
            for t in range(5):
                try:
                    us = timeit(lambda: fn(t))
                    row.append("t%d %6.1f us %5.1f TF" % (t, us, fl / us / 1e6))
                except Exception as e:
                    row.append("t%d failed" % t)
            print("C=%d %-4s %s [%dx%dx%d]  %s" % (C, name, form, M, N, K, " | ".join(row)))
