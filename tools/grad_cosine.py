"""Per-backward-stage cosine of the HIP gradient (and of the fp32 CPU oracle's) to an fp64 evaluation of the same graph.

    python tools/grad_cosine.py --init reference --batch 32 [--cache gpurun_out/gradref] [--tensors 12]
    MMFN_WINOGRAD_MIN_C=9999 python tools/grad_cosine.py ...      # A/B: every 3x3 convolution as a direct implicit GEMM

--init closed     the closed-form weight fill of the golden fixtures (oracle/fixtures.py)
--init reference  what bench.py trains from: torch.manual_seed(42) + the model class's own initialisation
                  (run_steps/utils.py:77-84 init_torch, model_vec.py:164-177 _init_weights, torchvision defaults)
The oracle gradients (fp32 and fp64, ~2 min at batch 32) are cached under --cache so that A/B runs share them."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--init", default="reference", choices=["reference", "closed"])
    ap.add_argument("--variant", default="vec")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--n-lidar", type=int, default=16384)
    ap.add_argument("--cache", default="")
    ap.add_argument("--tensors", type=int, default=10, help="worst tensors of the shallowest stage to list")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    args = ap.parse_args()

    import bench
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd import model as M
    from oracle import fixtures, gradcheck, harness

    torch.set_num_threads(bench.usable_cores())
    dev = "cuda:0"
    cfg = GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, act_dtype=args.dtype)
    cls = {"vec": M.MMFN, "img": M.MMFNImg, "rad": M.MMFNRad}[args.variant]
    if args.init == "reference":
        torch.manual_seed(42)
        net = cls(cfg, dev)
        oracle = harness.build_oracle(args.variant, dropout=0.0)
        oracle.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
    else:
        oracle = harness.build_oracle(args.variant, dropout=0.0)
        net = cls(cfg, dev)
        net.load_state_dict(oracle.state_dict(), strict=True)
    batch = fixtures.synthetic_batch(args.batch, args.variant, seed=42, n_lidar=args.n_lidar, lanes=64)
    fargs = harness.forward_args(batch, args.variant)
    cache = "%s_%s_%s_b%d.pt" % (args.cache, args.init, args.variant, args.batch) if args.cache else ""
    if cache and os.path.exists(cache):
        ref = torch.load(cache)
    else:
        t0 = time.time()
        loss32, g32, loss64, g64, _ = gradcheck.oracle_gradients(oracle, fargs, batch["gt_wp"])
        ref = {"loss32": float(loss32), "loss64": float(loss64), "g32": g32, "g64": g64}
        print("oracle fp32 + fp64 gradients: %.0f s" % (time.time() - t0), flush=True)
        if cache:
            os.makedirs(os.path.dirname(cache) or ".", exist_ok=True)
            torch.save(ref, cache)
    g32, g64 = ref["g32"], ref["g64"]

    net.train()
    eng = net._engine_for()
    to = lambda t: t.to(dev).contiguous()
    inp = {"rgb_u8": to(batch["rgb_u8"]), "lidar_pts": to(batch["lidar_pts"]), "lane": to(batch["lane"]),
           "lane_num": to(batch["lane_num"].to(torch.int32)), "target_point": to(batch["target_point"]), "velocity": to(batch["velocity"])}
    if args.variant == "img":
        inp["map"] = to(batch["map_u8"].float())
    if args.variant == "rad":
        inp["radar"], inp["radar_adj"] = to(batch["radar"]), to(batch["radar_adj"])
    _, loss = eng.forward(inp, True, batch["gt_wp"].to(dev))
    eng.backward()
    net._layout.attach_grads()
    torch.cuda.synchronize()
    hip = {n: p.grad for n, p in net.named_parameters()}
    tag = "%s init, %s B=%d, %s, MMFN_WINOGRAD_MIN_C=%s, MMFN_LAZY_BN=%s" % (
        args.init, args.variant, args.batch, args.dtype, os.environ.get("MMFN_WINOGRAD_MIN_C", "64 (default)"),
        os.environ.get("MMFN_LAZY_BN", "1 (default)"))
    print("[%s]" % tag)
    print("  loss: HIP %.7f  CPU fp32 %.7f  fp64 %.7f" % (loss.item(), ref["loss32"], ref["loss64"]))
    cos = gradcheck.stage_cosines(hip, g32, g64)
    for st, (h, c) in cos.items():
        print("  stage %d: cos(HIP, fp64) %.6f   cos(CPU fp32, fp64) %.6f   bar %.6f   1-cos ratio HIP/CPU %.2f"
              % (st, h, c, gradcheck.stage_bar(c), (1 - h) / max(1 - c, 1e-12)))
    rows = gradcheck.tensor_rows(hip, g32, g64)
    for st in sorted(cos):
        sel = sorted((r for r in rows if r[0] == st), key=lambda r: r[5])[:args.tensors if st == max(cos) else 3]
        for r in sel:
            print("    stage %d  %-58s |g64| %.3e  relerr HIP %.3e CPU %.3e  cos HIP %.6f CPU %.6f" % (r[0], r[1][-58:], r[2], r[3], r[4], r[5], r[6]))


if __name__ == "__main__":
    main()
