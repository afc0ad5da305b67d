"""Dev tool: the 36-batch Winograd-domain GEMMs of the step (forward NT, adjoint data gradient NN, weight gradient TN) per tile
shape, in isolation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
dev = "cuda:0"
iters = 20

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (T, C) in ((8192, 64), (2048, 128), (512, 256), (128, 512)):
    V = torch.randn(36, T, C, device=dev); U = torch.randn(36, C, C, device=dev); M = torch.empty(36, T, C, device=dev)
    dU = torch.empty(36, C, C, device=dev)
    fl = 2.0 * 36 * T * C * C
    for tile in (1, 2, 3, 4, 7):
        for sk in ((1,) if T > 128 else (1, 2, 4)):
            try:
                us = timeit(lambda: ops.gemm(V, U, M, T, C, C, C, C, C, ops.A_ROWMAJOR, ops.B_NK, batch=36, strideA=T * C, strideB=C * C, strideC=T * C, tile=tile, splitk=sk))
                print("fwd  NT  T=%4d C=%3d tile %d sk %d  %7.1f us  %6.1f TF/s" % (T, C, tile, sk, us, fl / us / 1e6))
                us = timeit(lambda: ops.gemm(V, U, M, T, C, C, C, C, C, ops.A_ROWMAJOR, ops.B_KN, batch=36, strideA=T * C, strideB=C * C, strideC=T * C, tile=tile, splitk=sk))
                print("adj  NN  T=%4d C=%3d tile %d sk %d  %7.1f us  %6.1f TF/s" % (T, C, tile, sk, us, fl / us / 1e6))
            except Exception as exc:
                print("tile", tile, "sk", sk, "failed", exc)
        for sk in (1, 2, 4):
            try:
                us = timeit(lambda: ops.gemm(M, V, dU, C, C, T, C, C, C, ops.A_COLMAJOR, ops.B_KN, batch=36, strideA=T * C, strideB=T * C, strideC=C * C, tile=tile, splitk=sk))
                print("wgrd TN  T=%4d C=%3d tile %d sk %d  %7.1f us  %6.1f TF/s" % (T, C, tile, sk, us, fl / us / 1e6))
            except Exception as exc:
                print("tile", tile, "sk", sk, "failed", exc)
