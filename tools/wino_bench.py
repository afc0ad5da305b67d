"""Dev tool: Winograd vs implicit-GEMM conv at the layer3 / layer4 shapes of the B=32 step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
dev = "cuda:0"

def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for H, C in ((64, 64), (32, 128), (16, 256), (8, 512)):
    x = torch.randn(32, H, H, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05; y = torch.empty_like(x)
    ops.WINOGRAD_MIN_CHANNELS = 0
    d = t(lambda: ops.conv2d_fwd(x, w, 1, 1, out=y))
    ops.WINOGRAD_MIN_CHANNELS = 64
    wi = t(lambda: ops.conv2d_fwd(x, w, 1, 1, out=y))
    print("%2dx%2d c%3d: implicit GEMM %6.1f us | winograd %6.1f us" % (H, H, C, d, wi))
