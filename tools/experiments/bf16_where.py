"""Where does a bf16 pipeline lose gradient direction on this network?  CPU oracle under torch.autocast(bfloat16) against itself in
fp32 (reference-style init, batch 8), with optional extra roundings that emulate choices of the HIP bf16 mode:
  --round-gpt-stream   the transformers' residual stream (and its gradient) rounded to bf16 after every block (autocast keeps it fp32)
  --round-fusion-add   the upsample-add of the transformer output into the trunk features rounded to bf16
python tools/experiments/bf16_where.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


class RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def main():
    import bench
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    from oracle import gradcheck, harness
    torch.set_num_threads(bench.usable_cores())
    B = int(os.environ.get("B", "8"))
    oracle = harness.build_oracle("vec", dropout=0.0)
    torch.manual_seed(42)
    oracle.load_state_dict({k: v.detach().cpu() for k, v in MMFN(GlobalConfig(), "cpu").state_dict().items()}, strict=True)
    inp, gt = bench.synth_inputs(B, torch.device("cpu"), seed=42)
    args = harness.forward_args(bench.oracle_batch_from_inputs(inp, "vec"), "vec")

    def grads(autocast, hooks=()):
        handles = [h() for h in hooks]
        oracle.train()
        for p in oracle.parameters():
            p.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            pred = oracle(*args)
        loss = harness.l1_waypoint_loss(pred.float(), gt)
        loss.backward()
        for hs in handles:
            for h in hs:
                h.remove()
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in oracle.named_parameters() if p.grad is not None}

    def gpt_stream():
        hs = []
        for i in range(1, 5):
            gpt = getattr(oracle.encoder, "transformer%d" % i)
            for blk in gpt.blocks:
                hs.append(blk.register_forward_hook(lambda m, i, o: RoundBoth.apply(o)))
            hs.append(gpt.drop.register_forward_hook(lambda m, i, o: RoundBoth.apply(o)))
        return hs

    def gpt_ln():
        hs = []
        for i in range(1, 5):
            gpt = getattr(oracle.encoder, "transformer%d" % i)
            for blk in gpt.blocks:
                for ln in (blk.ln1, blk.ln2):
                    hs.append(ln.register_forward_hook(lambda m, i, o: RoundBoth.apply(o)))
            hs.append(gpt.ln_f.register_forward_hook(lambda m, i, o: RoundBoth.apply(o)))
        return hs

    def fusion_add():   # the trunk features after "+ upsampled transformer output" (fp32 under autocast: the transformer output is fp32)
        hs = []
        enc = oracle.encoder
        for trunk in (enc.image_encoder.features, enc.lidar_encoder._model, enc.img_map_encoder.features):
            for li in (2, 3, 4):
                hs.append(getattr(trunk, "layer%d" % li).register_forward_pre_hook(lambda m, i: (RoundBoth.apply(i[0]),)))
        return hs

    l32, g32 = grads(False)
    rows = [("autocast", ()), ("autocast + bf16 transformer residual stream", (gpt_stream,)),
            ("autocast + bf16 LayerNorm outputs", (gpt_ln,)), ("autocast + both", (gpt_stream, gpt_ln)),
            ("autocast + stream + bf16 fusion add", (gpt_stream, fusion_add)), ("autocast + bf16 fusion add", (fusion_add,)),
            ("fp32 + bf16 transformer residual stream only", None)]
    for name, hooks in rows:
        if hooks is None:
            l, g = grads(False, (gpt_stream,))
        else:
            l, g = grads(True, hooks)
        cos = []
        for st in range(4):
            names = [k for k in g32 if gradcheck.stage_of(k) == st]
            a = torch.cat([g32[k].flatten().double() for k in names])
            c = torch.cat([g[k].flatten().double() for k in names])
            cos.append(float(torch.dot(a, c) / (a.norm() * c.norm())))
        print("%-50s loss %.6f (fp32 %.6f)  stage cosines %s" % (name, l, l32, "  ".join("%.4f" % c for c in cos)), flush=True)


if __name__ == "__main__":
    main()
