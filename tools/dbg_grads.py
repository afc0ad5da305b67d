"""Dev tool: per-parameter gradient error of the HIP path and of the fp32 CPU oracle vs an fp64 oracle."""
import copy, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from test_e2e_gpu import _setup, _dev_args, DEV
from oracle import harness
variant = sys.argv[1] if len(sys.argv) > 1 else 'vec'
oracle, net, batch, args = _setup(variant)
o64 = copy.deepcopy(oracle).double()
def to64(a):
    if torch.is_tensor(a): return a.double() if a.is_floating_point() else a
    if isinstance(a, (list, tuple)): return type(a)(to64(x) for x in a)
    return a
_, loss64, g64 = harness.train_step(o64, to64(args), batch["gt_wp"].double())
pred_ref, loss_ref, grads_ref = harness.train_step(oracle, args, batch["gt_wp"])
net.train()
pred = net(*_dev_args(args))
loss = torch.nn.functional.l1_loss(pred, batch["gt_wp"].to(DEV), reduction="none").mean()
loss.backward()
print("loss f64 %.9f cpu32 %.9f gpu %.9f" % (loss64.item(), loss_ref.item(), loss.item()))
rows = []
for name, p in net.named_parameters():
    if g64[name] is None: continue
    t = g64[name]; n = max(t.norm().item(), 1e-30)
    eg = (p.grad.detach().cpu().double() - t).norm().item() / n
    ec = (grads_ref[name].double() - t).norm().item() / n
    rows.append((name, eg, ec, n))
import math
gmax = max(r[3] for r in rows)
rows = [r for r in rows if r[3] > 1e-6 * gmax]
worst = sorted(rows, key=lambda r: -r[1] / max(r[2], 1e-7))[:40]
for r in worst: print("%-75s gpu %.2e cpu32 %.2e |g| %.2e" % r)
big = [r for r in rows if r[3] > 1e-6]
print("median gpu/cpu ratio (|g|>1e-6):", sorted(r[1] / max(r[2], 1e-9) for r in big)[len(big) // 2])
print("max gpu rel err (|g|>1e-6): %.3e ; max cpu32 rel err: %.3e" % (max(r[1] for r in big), max(r[2] for r in big)))
