"""Time the batched transformer weight-gradient launches (ops.linear_dw_batched / colsum_batched) against eight single ones."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmfn_amd import ops
dev = torch.device("cuda:0")
M, nb = 6144, 8


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for C in (64, 128, 256):
    for (N, K) in ((C, 4 * C), (4 * C, C), (C, C), (3 * C, C)):
        dy = torch.randn(nb, M, N, device=dev)
        x = torch.randn(nb, M, K, device=dev)
        stride = N * K + 4096
        out = torch.zeros(nb * stride, device=dev)
        out0 = out[:N * K].view(N, K)
        t_b = timeit(lambda: ops.linear_dw_batched(dy, x, out0, stride))
        ref = torch.stack([dy[i].t() @ x[i] for i in range(nb)])
        got = torch.stack([out[i * stride:i * stride + N * K].view(N, K) for i in range(nb)])
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        singles = [out[i * stride:i * stride + N * K].view(N, K) for i in range(nb)]
        t_s = timeit(lambda: [ops.linear_dw(dy[i], x[i], out=singles[i]) for i in range(nb)])
        print("C=%3d dW[%4d x %4d]  batched %7.1f us   8 singles %7.1f us   rel err %.1e" % (C, N, K, t_b, t_s, err))
    g = torch.randn(nb, M, 4 * C, device=dev)
    o = torch.zeros(nb * (4 * C + 64), device=dev)
    t_b = timeit(lambda: ops.colsum_batched(g, o[:4 * C], 4 * C + 64))
    got = torch.stack([o[i * (4 * C + 64):i * (4 * C + 64) + 4 * C] for i in range(nb)])
    err = (got - g.sum(1)).abs().max().item() / g.sum(1).abs().max().item()
    outs = [o[i * (4 * C + 64):i * (4 * C + 64) + 4 * C] for i in range(nb)]
    t_s = timeit(lambda: [ops.colsum(g[i], outs[i]) for i in range(nb)])
    print("C=%3d colsum[%d]          batched %7.1f us   8 singles %7.1f us   rel err %.1e" % (C, 4 * C, t_b, t_s, err))
