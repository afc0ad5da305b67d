import math, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mmfn_amd import ops
dev = "cuda:0"
B, T, NH = 16, 256, 4
rng = torch.tensor([5, 1], dtype=torch.int64, device=dev)
for HS in (16, 32, 64, 128):
    C = NH * HS
    qkv = torch.randn(B * T, 3 * C, device=dev); dO = torch.randn(B * T, C, device=dev)
    o = torch.empty(B * T, C, device=dev); lse = torch.empty(B, NH, T, device=dev)
    dqkv = torch.empty_like(qkv); delta = torch.empty(B, NH, T, device=dev)
    sc = 1 / math.sqrt(HS)
    fwd = lambda: ops.attention_fwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, C, lse, B, T, NH, HS, sc, drop_p=0.1, rng_state=rng, rng_stream=3)
    bwd = lambda: ops.attention_bwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, dO, C, lse, delta, dqkv[:, C:], dqkv, dqkv[:, 2 * C:], 3 * C, B, T, NH, HS, sc, drop_p=0.1, rng_state=rng, rng_stream=3)
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print("T=256 B=16 attn %s HS=%3d  %7.1f us" % (name, HS, e0.elapsed_time(e1) / 20 * 1e3))
