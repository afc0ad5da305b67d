"""Dev tool: time the fused attention kernels at the MMFN shapes (B=32, T=192, 4 heads)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
dev = "cuda:0"
B, T, NH = int(os.environ.get("ATTN_B", "32")), int(os.environ.get("ATTN_T", "192")), 4
iters = 20
DT = torch.bfloat16 if os.environ.get("ATTN_DTYPE") == "bf16" else torch.float32   # bf16: the bf16 mode's kernels (attention16.hip)
P = float(os.environ.get("ATTN_DROP", "0"))
rng = torch.tensor([5, 1], dtype=torch.int64, device=dev)
for HS in (16, 32, 64, 128):
    C = NH * HS
    qkv = torch.randn(B * T, 3 * C, device=dev).to(DT)
    dO = torch.randn(B * T, C, device=dev).to(DT)
    o = torch.empty(B * T, C, device=dev, dtype=DT); lse = torch.empty(B, NH, T, device=dev)
    dqkv = torch.empty_like(qkv); delta = torch.empty(B, NH, T, device=dev)
    sc = 1 / math.sqrt(HS)
    fwd = lambda: ops.attention_fwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, C, lse, B, T, NH, HS, sc, drop_p=P, rng_state=rng, rng_stream=3)
    bwd = lambda: ops.attention_bwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, dO, C, lse, delta, dqkv[:, C:], dqkv, dqkv[:, 2 * C:], 3 * C, B, T, NH, HS, sc,
                                    drop_p=P, rng_state=rng, rng_stream=3)
    for name, fn, mult in (("fwd", fwd, 1.0), ("bwd", bwd, 3.5)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 4.0 * T * T * C * B * mult
        print("attn %s HS=%3d  %7.1f us  %6.2f TF/s (algorithmic)" % (name, HS, ms * 1e3, fl / ms / 1e9))
