"""Dev tool: gradient quality of the bf16-operand mode against the fp32 HIP path at the bench size (B=32), per backward stage,
under a REALISTIC initialisation (the reference's own: torchvision kaiming-normal ResNets, N(0, 0.02) GPT linears, PyTorch
defaults elsewhere; seed 42 as run_steps/utils.py:77-84) and, for contrast, under the closed-form test fill.
Prints per stage: cosine(g_bf16, g_fp32), relative error, and the loss difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mmfn_amd.config import GlobalConfig
from mmfn_amd.model import MMFN

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))


def grads(net, inp, gt):
    eng = net._engine_for()
    net.train()
    _, loss = eng.forward(inp, True, gt)
    eng.backward()
    torch.cuda.synchronize()
    L = net._layout
    return float(loss.item()), L.grads[:L.tail].clone()


def report(tag, sd):
    inp, gt = bench.synth_inputs(B, dev, seed=42)
    out = {}
    for dtype in ("f32", "bf16"):
        torch.manual_seed(42)
        net = MMFN(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, gemm_dtype=dtype), dev)
        if sd is not None:
            net.load_state_dict(sd, strict=True)
        out[dtype] = grads(net, inp, gt) + (net._layout,)
    (l32, g32, L), (l16, g16, _) = out["f32"], out["bf16"]
    print("[%s] loss fp32 %.6f bf16 %.6f  rel diff %.2e" % (tag, l32, l16, abs(l32 - l16) / abs(l32)))
    for st, (b, e) in enumerate(L.stage_ranges):
        e = min(e, L.tail)
        a, c = g32[b:e].double(), g16[b:e].double()
        cos = float(torch.dot(a, c) / (a.norm() * c.norm()))
        print("   stage %d (%s): cosine %.5f  rel err %.3e  |g| %.3e" % (st, ["scale 4 + head", "scale 3", "scale 2", "scale 1 + stems + VectorNet"][st], cos,
                                                                          float((a - c).norm() / a.norm()), float(a.norm())))
    a, c = g32.double(), g16.double()
    print("   all: cosine %.5f rel err %.3e" % (float(torch.dot(a, c) / (a.norm() * c.norm())), float((a - c).norm() / a.norm())))


report("reference init (seed 42)", None)
from oracle import harness
report("closed-form test fill", harness.build_oracle("vec", dropout=0.0).state_dict())


def oracle_yardstick(Bo=8):
    """What torch's OWN mixed precision does to this network's gradients: the CPU oracle under torch.autocast(bfloat16)
    against itself in fp32, same reference-style init, per backward stage.  The yardstick for the HIP bf16 mode."""
    from oracle import fixtures, harness
    from mmfn_amd.params import FlatLayout
    torch.manual_seed(42)
    torch.set_num_threads(bench.usable_cores())
    net = MMFN(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0), "cpu")   # reference-style init (parameter skeleton)
    oracle = harness.build_oracle("vec", dropout=0.0)
    oracle.load_state_dict(net.state_dict(), strict=True)
    inp, gt = bench.synth_inputs(Bo, torch.device("cpu"), seed=42)
    args = harness.forward_args(bench.oracle_batch_from_inputs(inp, "vec"), "vec")

    def run(autocast):
        oracle.train()
        for p in oracle.parameters():
            p.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            pred = oracle(*args)
        loss = harness.l1_waypoint_loss(pred.float(), gt)
        loss.backward()
        return float(loss), {k: p.grad.detach().clone() for k, p in oracle.named_parameters() if p.grad is not None}

    l32, g32 = run(False)
    l16, g16 = run(True)
    print("[CPU oracle, torch.autocast(bfloat16) vs fp32, batch %d] loss %.6f vs %.6f" % (Bo, l16, l32))
    for st in range(4):
        names = [k for k in g32 if FlatLayout.stage_of(k) == st]
        a = torch.cat([g32[k].flatten().double() for k in names]); c = torch.cat([g16[k].flatten().double() for k in names])
        print("   stage %d: cosine %.5f  rel err %.3e" % (st, float(torch.dot(a, c) / (a.norm() * c.norm())), float((a - c).norm() / a.norm())))


if os.environ.get("ORACLE_YARDSTICK", "1") == "1":
    oracle_yardstick()
