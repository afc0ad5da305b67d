"""How long do the small transformer GEMMs take in a dependent chain (graph of 50 back-to-back launches), per tile / split choice?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmfn_amd import ops
dev = torch.device("cuda:0")
M = 6144


def chain_time(fn, n=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 / n * 1e6


for (N, K) in ((192, 64), (64, 64), (256, 64), (64, 256), (384, 128), (128, 128), (512, 128), (128, 512)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
    b = torch.randn(N, device=dev)
    out = []
    for tile in (2, 3, 4, 1):
        for sk in (1, 2, 4):
            if sk > 1 and K // 16 // sk < 1:
                continue
            try:
                t = chain_time(lambda: ops.linear_fwd(x, w, b, out=y, tile=tile, splitk=sk))
                out.append((t, tile, sk))
            except Exception as e:
                pass
    t_auto = chain_time(lambda: ops.linear_fwd(x, w, b, out=y))
    out.sort()
    print("fwd %dx%dx%d  table/auto %5.1f us | best %s" % (M, N, K, t_auto, "  ".join("t%d sk%d %.1f" % (tl, sk, t) for t, tl, sk in out[:4])))
