"""Dev tool: the fused GPT-block kernels against the chain of separate kernels they replace, B = 32, T = 192, C = 64 / 128."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from mmfn_amd import ops
import test_gpt_block_gpu as tg

dev = torch.device("cuda:0")
B, T, NH = int(os.environ.get("GPT_B", "32")), 192, 4
M = B * T
rng = torch.tensor([11, 3], dtype=torch.int64, device=dev)


def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for C in (64, 128):
    HS = C // NH
    g = torch.Generator().manual_seed(C)
    p = tg._params(C, g, dev)
    x = torch.randn(M, C, generator=g).to(dev)
    out = tg._bufs(B, T, C, NH, dev)
    d = ops.gpt_block_desc(B, T, C, NH, attn_pdrop=0.1, resid_pdrop=0.1, rng_state=rng, rng_stream=40, x=x, **p, **out)
    t_attn = timeit(lambda: ops.gpt_block_attn_fwd(d))
    t_mlp = timeit(lambda: ops.gpt_block_mlp_fwd(d))
    t_both = timeit(lambda: (ops.gpt_block_attn_fwd(d), ops.gpt_block_mlp_fwd(d)))
    t_plain = timeit(lambda: tg._unfused_forward(ops, p, x, B, T, C, NH, out, 0.1, 0.1, rng, 40))
    print("C=%3d fwd: fused attn %.1f us + mlp %.1f us (pair %.1f)   separate kernels %.1f us" % (C, t_attn, t_mlp, t_both, t_plain))
    # backward rows
    e = lambda *s: torch.zeros(s, device=dev)
    nrow = M // ops.GPT_ROWS
    dqkv, g1u = torch.randn(M, 3 * C, device=dev), torch.randn(M, C, device=dev)
    o = dict(g_below=e(M, C), gd_below=e(M, C), part_ln1=e(nrow, 3, C), gh=e(M, 4 * C), g1=e(M, C), gd2=e(M, C), go=e(M, C), part_ln2=e(nrow, 3, C))
    up = ops.gpt_block_desc(B, T, C, NH, resid_pdrop=0.1, rng_state=rng, rng_stream=50, rng_stream_below=47, below_colsum=True, x=x,
                            mu1=out["mu1"], rs1=out["rs1"], dqkv=dqkv, g1=g1u, g_below=o["g_below"], gd_below=o["gd_below"],
                            part_ln1=o["part_ln1"], **p)
    lo = ops.gpt_block_desc(B, T, C, NH, resid_pdrop=0.1, rng_state=rng, rng_stream=47, x1=out["x1"], mu2=out["mu2"], rs2=out["rs2"],
                            h=out["h"], g=o["g_below"], gd=o["gd_below"], gh=o["gh"], g1=o["g1"], gd2=o["gd2"], go=o["go"],
                            part_ln2=o["part_ln2"], **p)
    t_rows = timeit(lambda: ops.gpt_block_bwd_rows(up, lo))
    gw, gb, cs = e(C), e(C), e(C)
    ga, ga2 = e(M, C), e(M, C)

    def plain_bwd():
        ops.linear_dx(dqkv, p["wqkv"], out=ga)
        ops.layernorm_bwd(ga, x, p["ln1_w"], p["ln1_b"], out["mu1"], out["rs1"], o["g_below"], gw, gb, 0, dres=g1u, dx_dropped=o["gd_below"],
                          drop_p=0.1, rng_state=rng, rng_stream=49, dx_colsum=cs)
        ops.linear_dx(o["gd_below"], p["w2"], out=o["gh"], aux=out["h"], ldaux=4 * C)
        ops.linear_dx(o["gh"], p["w1"], out=ga2)
        ops.layernorm_bwd(ga2, out["x1"], p["ln2_w"], p["ln2_b"], out["mu2"], out["rs2"], o["g1"], gw, gb, 0, dres=o["g_below"],
                          dx_dropped=o["gd2"], drop_p=0.1, rng_state=rng, rng_stream=48, dx_colsum=cs)
        ops.linear_dx(o["gd2"], p["wproj"], out=o["go"])
    t_pb = timeit(plain_bwd)
    print("C=%3d bwd rows: fused %.1f us   separate kernels (6 + 2 finalize) %.1f us" % (C, t_rows, t_pb))
