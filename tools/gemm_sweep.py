"""Dev tool: exhaustive tile x split-K sweep on the step's dominant shapes vs the automatic choice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
dev = "cuda:0"; B = 32; iters = 12

def t(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def sweep(name, make, fl, sks):
    auto = t(make(0, 0))
    res = []
    for tile in (1, 2, 3, 4):
        for sk in sks:
            try:
                res.append((t(make(tile, sk)), tile, sk))
            except Exception as e:
                pass
    res.sort()
    best = res[0]
    print("%-34s auto %7.1f us %6.1f TF/s | best t%d sk%-3d %7.1f us %6.1f TF/s (%+.1f%%) | next: %s" % (
        name, auto, fl / auto / 1e6, best[1], best[2], best[0], fl / best[0] / 1e6, (auto / best[0] - 1) * 100,
        " ".join("t%d/sk%d:%.0f" % (b[1], b[2], b[0]) for b in res[1:4])))

for H, C in ((64, 64), (32, 128), (16, 256), (8, 512)):
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    y = torch.empty(B, H, H, C, device=dev); dy = torch.randn(B, H, H, C, device=dev); dw = torch.empty_like(w)
    fl = 2.0 * B * H * H * C * 9 * C
    sweep("conv  %dx%d c%d" % (H, H, C), lambda tile, sk: (lambda: ops.conv2d_fwd(x, w, 1, 1, out=y, tile=tile, splitk=sk)), fl, (1, 2, 3, 4, 6))
    sweep("wgrad %dx%d c%d" % (H, H, C), lambda tile, sk: (lambda: ops.conv2d_wgrad(dy, x, tuple(w.shape), 1, 1, out=dw, tile=tile, splitk=sk)), fl,
          (2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96))
for C in (512, 256, 128):
    M = 6144
    for (N, K) in ((4 * C, C), (C, 4 * C), (3 * C, C), (C, C)):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); dy = torch.randn(M, N, device=dev)
        y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
        fl = 2.0 * M * N * K
        sweep("lin fwd %dx%dx%d" % (M, N, K), lambda tile, sk: (lambda: ops.linear_fwd(x, w, out=y, tile=tile, splitk=sk)), fl, (1, 2, 4))
        sweep("lin dx  %dx%dx%d" % (M, K, N), lambda tile, sk: (lambda: ops.linear_dx(dy, w, out=dx, tile=tile, splitk=sk)), fl, (1, 2, 4))
        sweep("lin dw  %dx%dx%d" % (N, K, M), lambda tile, sk: (lambda: ops.linear_dw(dy, x, out=dw, tile=tile, splitk=sk)), fl, (1, 2, 3, 4, 6, 8, 12, 16, 24, 32))
