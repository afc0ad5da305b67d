import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from mmfn_amd import ops
import test_gpt_block_gpu as tg
dev = torch.device("cuda:0")
B, T, NH = 32, 192, 4
for C in (64, 128):
    g = torch.Generator().manual_seed(C)
    p = tg._params(C, g, dev)
    x = torch.randn(B * T, C, generator=g).to(dev)
    ref = tg._ref_block(p, x, B, T, C, NH)
    fused, plain = tg._bufs(B, T, C, NH, dev), tg._bufs(B, T, C, NH, dev)
    d = ops.gpt_block_desc(B, T, C, NH, x=x, **p, **fused)
    ops.gpt_block_attn_fwd(d); ops.gpt_block_mlp_fwd(d)
    tg._unfused_forward(ops, p, x, B, T, C, NH, plain, 0.0, 0.0, None, 0)
    torch.cuda.synchronize()
    for name in ("a", "qkv", "o", "x1", "a2", "h", "x2"):
        r = ref[name].to(dev)
        ef = ((fused[name].double() - r) ** 2).mean().sqrt().item()
        ep = ((plain[name].double() - r) ** 2).mean().sqrt().item()
        print("C=%d %-4s rms err vs fp64: fused %.3e  separate %.3e  (rms %.3e)" % (C, name, ef, ep, (r ** 2).mean().sqrt().item()))
