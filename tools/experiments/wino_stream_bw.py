"""Achieved HBM bandwidth of the Winograd streaming kernels (input / output+stats / outgrad+BN / adjoint / bn_apply / col_partial)
at the four layer shapes of the B=32 step: are they at the HBM roofline?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmfn_amd import ops
from mmfn_amd._lib import lib, ptr
import ctypes
dev = torch.device("cuda:0")
B = 32


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


L = lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for H, C in ((64, 64), (32, 128), (16, 256), (8, 512)):
    A = B * H * H * C * 4 / 1e6   # activation MB
    T = B * (H // 4) * (H // 4)
    x = torch.randn(B, H, H, C, device=dev)
    V = torch.empty(36 * T * C, device=dev)
    y = torch.empty_like(x)
    g = torch.randn_like(x)
    mean, rstd, w, means = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev), torch.zeros(2, C, device=dev)
    ws = ops.norm_workspace(dev)
    nblk = ctypes.c_int(0)
    rows = []
    t = timeit(lambda: L.mmfn_wino_input_f32(ptr(x), ptr(V), B, H, H, C, 4, st()))
    rows.append(("wino4_input", t, A * 3.25))
    t = timeit(lambda: L.mmfn_wino_output_stats_f32(ptr(V), ptr(y), ptr(ws), ctypes.cast(ctypes.pointer(nblk), ctypes.c_void_p), B, H, H, C, st()))
    rows.append(("wino4_output_stats", t, A * 3.25))
    t = timeit(lambda: L.mmfn_wino_outgrad_bn_f32(ptr(g), ptr(y), ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(means), None, ptr(V), B, H, H, C, st()))
    rows.append(("wino4_outgrad_bn", t, A * 5.25))
    t = timeit(lambda: L.mmfn_wino_input_adjoint_f32(ptr(V), None, ptr(y), B, H, H, C, st()))
    rows.append(("wino4_input_adjoint", t, A * 3.25))
    M = B * H * H
    t = timeit(lambda: ops.bn_apply(x.view(M, C), y.view(M, C), mean, rstd, w, w, True))
    rows.append(("bn_apply", t, A * 2))
    dw, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    t = timeit(lambda: ops.bn_bwd_reduce(g.view(M, C), y.view(M, C), x.view(M, C), mean, rstd, dw, db, means))
    rows.append(("bn_bwd_reduce (2 launches)", t, A * 3))
    print("B=%d %dx%d c%d (activation %.1f MB):" % (B, H, H, C, A))
    for name, t, mb in rows:
        print("   %-28s %7.1f us  %6.0f MB  %5.2f TB/s" % (name, t, mb, mb / t))
