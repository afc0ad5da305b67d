"""Dev tool: time the sensor-ingest kernels (camera crop/normalise, LiDAR BEV splat) and report achieved HBM GB/s against
their ALGORITHMIC bytes (every input byte the kernel needs read once, every output byte written once).

  python tools/ingest_bench.py            # HIP-event timing, prints one line per case
  rocprofv3 --pmc FETCH_SIZE ... -- python tools/ingest_bench.py --once    # counter runs (tools/pmc_ingest.sh)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops

dev = "cuda:0"
once = "--once" in sys.argv
iters = 1 if once else 50


def timed(fn):
    for _ in range(0 if once else 5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


g = torch.Generator().manual_seed(0)
for B, N in ((32, 16384), (16, 65536), (32, 65536)):
    rgb = torch.randint(0, 256, (B, 300, 400, 3), generator=g, dtype=torch.uint8).to(dev)
    out = torch.empty(B, 256, 256, 3, device=dev)
    us = timed(lambda: ops.ingest_rgb_u8(rgb, out))
    nbytes = B * 256 * 256 * 3 * (1 + 4)
    print("ingest_rgb_u8   B=%2d            %7.1f us  %7.1f GB/s  (algorithmic %.1f MB: cropped u8 in + f32 NHWC out)"
          % (B, us, nbytes / us / 1e3, nbytes / 1e6))
    pts = torch.empty(B, N, 4)
    pts[..., 0:2] = torch.rand(B, N, 2, generator=g) * 40.0 - 20.0
    pts[..., 2] = torch.rand(B, N, generator=g) * 4.0 - 3.0
    pts[..., 3] = torch.rand(B, N, generator=g)
    pts = pts.to(dev)
    bev = torch.empty(B, 256, 256, 2, device=dev)
    us = timed(lambda: ops.lidar_splat(pts, bev))
    nbytes = B * N * 16 + B * 256 * 256 * 2 * 4
    print("lidar_splat     B=%2d N=%6d   %7.1f us  %7.1f GB/s  (algorithmic %.1f MB: XYZI points in + f32 BEV out)"
          % (B, N, us, nbytes / us / 1e3, nbytes / 1e6))
