"""Blocks per CU capped by extra dynamic LDS (MMFN_GEMM_DYN_LDS): the big transformer GEMMs with 128x128 / 128x64 / 64x64 tiles.
Needs a build of gemm_f32.hip with -DMMFN_GEMM_EXPERIMENTS (MMFN_EXTRA_FLAGS=-DMMFN_GEMM_EXPERIMENTS python -m mmfn_amd.build --force
into a scratch copy, or MMFN_HIP_LIB); result of round 5: profiles/r05_gemm_cap_sweep.txt (slower at every cap)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmfn_amd import ops
dev = "cuda:0"
M = 6144
def timeit(fn, iters=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (N, K, name) in ((2048, 512, "fc1"), (1536, 512, "qkv"), (512, 2048, "fc2"), (512, 512, "proj")):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev)
    fl = 2.0 * M * N * K
    row = []
    for tile in (1, 2, 3, 4, 5):
        row.append("t%d %6.1f/%6.1f" % (tile, timeit(lambda: ops.linear_fwd(x, w, b, out=y, tile=tile, splitk=1)), timeit(lambda: ops.linear_dx(dy, w, out=dx, tile=tile, splitk=1))))
    print("dynLDS %6s  %-4s fwd/dx us: %s" % (os.environ.get("MMFN_GEMM_DYN_LDS", "0"), name, "  ".join(row)), flush=True)
