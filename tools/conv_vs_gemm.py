"""Dev tool: implicit-GEMM conv vs a plain GEMM of the same M x N x K, at the step's size and at 4x the rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
dev = "cuda:0"

def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for B in (32, 128):
    for H, C in ((64, 64), (32, 128), (16, 256), (8, 512)):
        x = torch.randn(B, H, H, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05
        y = torch.empty(B, H, H, C, device=dev)
        M, N, K = B * H * H, C, 9 * C
        a = torch.randn(M, K, device=dev); wm = torch.randn(N, K, device=dev); ym = torch.empty(M, N, device=dev)
        fl = 2.0 * M * N * K
        best = {}
        for name, fn in (("conv", lambda tile, sk: ops.conv2d_fwd(x, w, 1, 1, out=y, tile=tile, splitk=sk)),
                         ("gemm", lambda tile, sk: ops.linear_fwd(a, wm, out=ym, tile=tile, splitk=sk))):
            res = []
            for tile in (1, 2, 3, 4):
                for sk in (1, 2, 3, 4, 6):
                    if sk > 1 and B * H * H * C // (64 * 64) * sk > 8192: continue
                    res.append((t(lambda: fn(tile, sk)), tile, sk))
            best[name] = min(res)
        print("B=%3d %2dx%2d c%3d  M=%6d N=%3d K=%4d | conv best t%d/sk%d %7.1f us %6.1f TF/s | plain GEMM best t%d/sk%d %7.1f us %6.1f TF/s" % (
            B, H, H, C, M, N, K, best["conv"][1], best["conv"][2], best["conv"][0], fl / best["conv"][0] / 1e6,
            best["gemm"][1], best["gemm"][2], best["gemm"][0], fl / best["gemm"][0] / 1e6))
