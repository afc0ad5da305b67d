"""bf16 training mode (GlobalConfig(act_dtype="bf16")) against the fp32 HIP path on the same weights and batch: loss, eval
waypoints, per-backward-stage gradient cosine / relative error, and a few optimizer steps of both.
  python tools/bf16_quality.py [--batch 8] [--oracle-init]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stage_stats(La, Lb):
    out = []
    for st, (b, e) in enumerate(La.stage_ranges):
        a, c = La.grads[b:min(e, La.tail)].double(), Lb.grads[b:min(e, Lb.tail)].double()
        cos = float((a * c).sum() / (a.norm() * c.norm() + 1e-300))
        rel = float((a - c).norm() / (a.norm() + 1e-300))
        out.append((st, cos, rel))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--variant", default="vec")
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()
    import bench
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN, MMFNImg
    dev = torch.device("cuda", 0)
    cls = {"vec": MMFN, "img": MMFNImg}[args.variant]
    torch.manual_seed(args.seed)
    a = cls(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0), dev)
    b = cls(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, act_dtype="bf16"), dev)
    b.load_state_dict(a.state_dict())
    a.train(), b.train()
    inp, gt = bench.synth_inputs(args.batch, dev, seed=args.seed, variant=args.variant)
    ea, eb = a._engine_for(), b._engine_for()
    _, la = ea.forward(inp, True, gt)
    ea.backward()
    _, lb = eb.forward(inp, True, gt)
    eb.backward()
    torch.cuda.synchronize()
    print("loss f32 %.6f  bf16 %.6f  rel %.2e" % (la.item(), lb.item(), abs(la.item() - lb.item()) / abs(la.item())))
    for st, cos, rel in stage_stats(a._layout, b._layout):
        print("backward stage %d: gradient cosine %.4f  relative error %.3f" % (st, cos, rel))
    for k in ("stage1", "gpt1", "gpt2", "gpt3", "gpt4"):
        ta, tb = ea.taps[k], eb.taps[k]
        if isinstance(ta, tuple):
            for m, (x, y) in enumerate(zip(ta, tb)):
                print("tap %s.%d rel err %.2e" % (k, m, float((x.float() - y.float()).norm() / x.float().norm())))
        else:
            print("tap %s rel err %.2e" % (k, float((ta.float() - tb.float()).norm() / ta.float().norm())))
    a.eval(), b.eval()
    with torch.no_grad():
        pa, _ = ea.forward(inp, False, None)
        pb, _ = eb.forward(inp, False, None)
    print("eval waypoints max |diff| %.3e (scale %.3f)" % (float((pa - pb).abs().max()), float(pa.abs().max())))
    a.train(), b.train()
    for i in range(5):
        la = a.train_step(inp, gt)
        lb = b.train_step(inp, gt)
    torch.cuda.synchronize()
    print("after 5 AdamW steps on the same batch: loss f32 %.5f bf16 %.5f" % (la.item(), lb.item()))


if __name__ == "__main__":
    main()
