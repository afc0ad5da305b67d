"""Per-shape timing of the LDS-resident-patch 3x3 convolution (mmfn_conv3x3_halo_bf16) against the two launches it replaces
(mmfn_bn_apply_bf16 -> mmfn_gemm_bf16 MMFN_G16_CONV_FWD with statistics) on the trunk shapes of the benched batch, over every
(tile, stages) the kernel offers.  --write stores the winners in mmfn_amd/tuning/gfx950_halo.json.
  python tools/halo_bench.py [--batch 32] [--write]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mmfn_amd import ops, ops16  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def timeit(fn, reps=20, rounds=3):
    for _ in range(3):
        fn()
    best = None
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e3
        best = t if best is None else min(best, t)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--write", action="store_true")
    a = ap.parse_args()
    B = a.batch
    table = {}
    for (H, C) in ((64, 64), (32, 128), (16, 256), (8, 512)):
        M = B * H * H
        g = torch.Generator().manual_seed(1)
        co = (torch.randn(B, H, H, C, generator=g) * 2).to(DEV).to(BF)
        res = torch.randn(B, H, H, C, generator=g).to(DEV).to(BF)
        w = (torch.randn(C, 3, 3, C, generator=g) * 0.05).to(DEV).to(BF)
        mean, rstd = torch.randn(C, device=DEV) * 0.3, torch.rand(C, device=DEV) + 0.5
        gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.2
        y = torch.empty_like(co)
        out = torch.empty(B, H, H, C, dtype=BF, device=DEV)
        stats = torch.zeros(2 * M // 64, 2, C, dtype=torch.float64, device=DEV)
        flops = 2.0 * M * 9 * C * C

        def old():
            ops.bn_apply(co.view(M, C), y.view(M, C), mean, rstd, gamma, beta, True, res=res.view(M, C))
            ops16.conv2d_fwd(y, w, 1, 1, out, stats=stats)

        def old_conv():
            ops16.conv2d_fwd(y, w, 1, 1, out, stats=stats)

        t_old, t_conv = timeit(old), timeit(old_conv)
        print("%dx%d C=%d  M=%d  %.2f GF:  apply + implicit GEMM %.1f us (GEMM alone %.1f us = %.0f TF/s)" % (H, H, C, M, flops / 1e9, t_old, t_conv,
                                                                                                  flops / t_conv / 1e6))
        best = None
        for pro in (1, 0):
            for tile in (1, 2, 3, 4):
                if tile in (2, 4) and C % 128:
                    continue
                for stages in (2, 3, 4):
                    def new():
                        if pro:
                            ops16.conv3x3_halo(co, w, out, stats=stats, bn_apply=(mean, rstd, gamma, beta, res, True, y), tile=tile, stages=stages)
                        else:
                            ops16.conv3x3_halo(y, w, out, stats=stats, tile=tile, stages=stages)
                    try:
                        t = timeit(new)
                    except Exception as e:   # a (tile, stages) whose LDS does not fit
                        print("    pro %d tile %d stages %d: %s" % (pro, tile, stages, str(e)[:60]))
                        continue
                    print("    pro %d tile %d stages %d: %6.1f us  %5.0f TF/s   x%.2f" % (pro, tile, stages, t, flops / t / 1e6, (t_old if pro else t_conv) / t))
                    if pro == 1 and (best is None or t < best[0]):
                        best = (t, tile, stages)
            if pro == 1 and best:
                for p in (0, 1, 2):
                    table["%d,%d,%d,%d,%d,%d" % (B, H, H, C, C, p)] = [best[1], best[2]]
                print("  best pro 1: tile %d stages %d %.1f us" % (best[1], best[2], best[0]))
    if a.write:
        path = ops16._HALO_TUNE_FILE
        old_t = {}
        if os.path.isfile(path):
            old_t = json.load(open(path))
        old_t.update(table)
        json.dump(old_t, open(path, "w"), indent=0, sort_keys=True)
        print("wrote", path)


if __name__ == "__main__":
    main()
