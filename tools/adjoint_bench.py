"""Dev tool: time the Winograd adjoint data-gradient transform (and the forward output transform for comparison)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
dev = "cuda:0"
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B = 32
for H, C in ((32, 128), (16, 256), (8, 512)):
    T = B * (H // 4) ** 2
    dV = torch.randn(36 * T * C, device=dev); dx = torch.empty(B, H, H, C, device=dev)
    us = timeit(lambda: ops._call("mmfn_wino_input_adjoint_f32", ops.ptr(dV), None, ops.ptr(dx), B, H, H, C, ops.stream()))
    us2 = timeit(lambda: ops._call("mmfn_wino_output_f32", ops.ptr(dV), None, ops.ptr(dx), B, H, H, C, 4, ops.stream()))
    print("H=%2d C=%3d  adjoint %6.1f us   (output transform %6.1f us)   %.1f MB" % (H, C, us, us2, (36 * T * C + B * H * H * C) * 4 / 1e6))
