"""Where does a bf16 trunk convolution's time go?  For the layer shapes of the B = 32 step: the convolution (implicit-GEMM gather)
against the plain NT GEMM of the same M, N, K, over tile and LDS-stage choices, isolated (back-to-back launches, events).
Run on the GPU box: python tools/experiments/conv16_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmfn_amd import ops, ops16  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


def main():
    tiles = {2: "64x64", 4: "64x128", 3: "128x64", 1: "128x128"}
    for (B, H, C) in ((32, 64, 64), (32, 32, 128), (32, 16, 256), (32, 8, 512)):
        x = torch.randn(B, H, H, C, device=DEV).to(BF)
        w = (torch.randn(C, 3, 3, C, device=DEV) * 0.05).to(BF)
        y = torch.empty(B, H, H, C, dtype=BF, device=DEV)
        M, N, K = B * H * H, C, 9 * C
        a2 = torch.randn(M, K, device=DEV).to(BF)
        w2 = w.view(C, K)
        gf = 2.0 * M * N * K / 1e9
        print("conv %dx%d c%d: M %d N %d K %d, %.1f GFLOP; 2.5 PF floor %.1f us; us conv / plain GEMM per (tile, stages)" % (H, H, C, M, N, K, gf, gf / 2500 * 1e3))
        for t, tn in tiles.items():
            row = []
            for st in (2, 3, 4):
                try:
                    tc = timed(lambda: ops16.conv2d_fwd(x, w, 1, 1, y, tile=t, stages=st))
                    tg = timed(lambda: ops16.gemm16(ops16.G16_NT, a2, w2, y, M, N, K, K, K, N, tile=t, stages=st))
                    row.append("s%d %5.1f/%5.1f" % (st, tc, tg))
                except Exception as exc:   # a (tile, stages) pair the kernel does not build
                    row.append("s%d  n/a      " % st)
            print("   %-8s %s" % (tn, " | ".join(row)))


if __name__ == "__main__":
    main()
