"""Dev tool: time representative GEMM / conv shapes of the MMFN step (B=32) in isolation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
dev = "cuda:0"
B = 32
iters = int(os.environ.get("ITERS", "20"))
only = os.environ.get("ONLY")

def timeit(fn, flops, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-44s %8.3f ms  %7.2f TF/s" % (name, ms, flops / ms / 1e9))

def conv_case(H, Cin, Cout, k, s, tile=0, splitk=0):
    p = k // 2
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    g, oshape = ops.conv_geom(x.shape, w.shape, s, p)
    y = torch.empty(oshape, device=dev); dy = torch.randn(oshape, device=dev)
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * oshape[0] * oshape[1] * oshape[2] * Cout * k * k * Cin
    tag = "%dx%d c%d->%d k%d s%d t%d sk%d" % (H, H, Cin, Cout, k, s, tile, splitk)
    if only in (None, "fwd"): timeit(lambda: ops.conv2d_fwd(x, w, s, p, out=y, tile=tile, splitk=splitk), fl, "conv  " + tag)
    if only in (None, "dgrad") and Cin % 4 == 0: timeit(lambda: ops.conv2d_dgrad(dy, w, tuple(x.shape), s, p, out=dx, tile=tile, splitk=splitk), fl, "dgrad " + tag)
    if only in (None, "wgrad"): timeit(lambda: ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, out=dw, tile=tile, splitk=splitk), fl, "wgrad " + tag)

def lin_case(M, N, K, tile=0, splitk=0):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); dy = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
    fl = 2.0 * M * N * K
    tag = "%dx%dx%d t%d sk%d" % (M, N, K, tile, splitk)
    if only in (None, "fwd"): timeit(lambda: ops.linear_fwd(x, w, out=y, tile=tile, splitk=splitk), fl, "lin fwd " + tag)
    if only in (None, "dgrad"): timeit(lambda: ops.linear_dx(dy, w, out=dx, tile=tile, splitk=splitk), fl, "lin dx  " + tag)
    if only in (None, "wgrad"): timeit(lambda: ops.linear_dw(dy, x, out=dw, tile=tile, splitk=splitk), fl, "lin dw  " + tag)

cases = os.environ.get("CASES", "all")
if cases in ("all", "conv"):
    for tile in (1, 2):
        conv_case(64, 64, 64, 3, 1, tile); conv_case(32, 128, 128, 3, 1, tile); conv_case(16, 256, 256, 3, 1, tile)
        conv_case(8, 512, 512, 3, 1, tile)
    for sk in (1, 2, 4, 8):
        conv_case(8, 512, 512, 3, 1, 2, sk)
    conv_case(64, 64, 128, 3, 2, 2); conv_case(16, 256, 512, 3, 2, 2)
if cases in ("all", "lin"):
    for tile in (1, 2):
        lin_case(6144, 2048, 512, tile); lin_case(6144, 512, 2048, tile); lin_case(6144, 1536, 512, tile); lin_case(6144, 512, 512, tile)
        lin_case(6144, 256, 256, tile); lin_case(6144, 64, 64, tile)
