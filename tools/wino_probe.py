"""Dev probe: rate of the batched GEMM stage a Winograd F(2x2,3x3) convolution would need (16 x [tiles x Cout x Cin])."""
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

for (T, C) in ((32768, 64), (8192, 128), (2048, 256), (512, 512)):
    V = torch.randn(16, T, C, device=dev); U = torch.randn(16, C, C, device=dev); M = torch.empty(16, T, C, device=dev)
    fl = 2.0 * 16 * T * C * C
    best = None
    for tile in (1, 2, 3, 4):
        us = t(lambda: ops.gemm(V, U, M, T, C, C, C, C, C, ops.A_ROWMAJOR, ops.B_NK, batch=16, strideA=T * C, strideB=C * C, strideC=T * C, tile=tile, splitk=1))
        if best is None or us < best[0]: best = (us, tile)
    direct = 2.0 * T * 4 * C * 9 * C
    print("tiles=%6d C=%3d: batched GEMM best t%d %7.1f us %6.1f TF/s  (direct conv = %.2f GF -> equivalent %6.1f TF/s before transforms)" % (
        T, C, best[1], best[0], fl / best[0] / 1e6, direct / 1e9, direct / best[0] / 1e6))
