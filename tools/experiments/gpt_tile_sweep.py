"""All four tiles (and the table's own choice, tile 0) for the transformer GEMM shapes of the B = 32 step, in isolation,
with the step's epilogues (bias / ReLU mask / residual + dropout); prints the table's pick next to the best."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmfn_amd import ops
dev = "cuda:0"
M = 6144
iters = 40
rng = torch.tensor([5, 1], dtype=torch.int64, device=dev)

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for C in [int(c) for c in os.environ.get("SWEEP_C", "512,256,128").split(",")]:
    for name, N, K in (("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev); y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
        res = torch.randn(M, N, device=dev); aux = torch.randn(M, K, device=dev)
        for kind in ("fwd", "dx", "dw"):
            row = []
            for tile in (0, 1, 2, 3, 4, 5, 6, 7):
                try:
                    if kind == "fwd":
                        if name in ("proj", "fc2"):
                            fn = lambda: ops.linear_fwd(x, w, b, out=y, tile=tile, res=res, ldr=N, drop_p=0.1, rng_state=rng, rng_stream=3)
                        elif name == "fc1":
                            fn = lambda: ops.linear_fwd(x, w, b, out=y, tile=tile, relu=True)
                        else:
                            fn = lambda: ops.linear_fwd(x, w, b, out=y, tile=tile)
                    elif kind == "dx":
                        if name == "fc2":
                            fn = lambda: ops.linear_dx(dy, w, out=dx, tile=tile, aux=aux, ldaux=K)
                        else:
                            fn = lambda: ops.linear_dx(dy, w, out=dx, tile=tile)
                    else:
                        fn = lambda: ops.linear_dw(dy, x, out=dw, tile=tile)
                    row.append(timeit(fn))
                except Exception as e:
                    row.append(float("nan"))
            best = min(range(1, 8), key=lambda t: row[t] if row[t] == row[t] else 1e30)
            fl = 2.0 * M * N * K
            print("C=%3d %-4s %-3s  table %6.1f us (%5.1f TF/s) | t1 %6.1f  t2 %6.1f  t3 %6.1f  t4 %6.1f  t5 %6.1f  t6 %6.1f  t7 %6.1f | best t%d %5.1f TF/s %s" % (
                C, name, kind, row[0], fl / row[0] / 1e6, row[1], row[2], row[3], row[4], row[5], row[6], row[7], best, fl / row[best] / 1e6, "<-- %.0f%%" % (100 * (row[0] - row[best]) / row[0]) if row[best] < 0.97 * row[0] else ""))
