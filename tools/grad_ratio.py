"""Dev tool: distribution of |hip - f64| / |cpu_fp32 - f64| gradient error ratios (see tests/test_e2e_gpu.py)."""
import copy, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_e2e_gpu as T
from oracle import harness
variant = sys.argv[1] if len(sys.argv) > 1 else "vec"
oracle, net, batch, args = T._setup(variant)
seed = int(os.environ.get("SEED", "42"))
if seed != 42:
    from oracle import fixtures
    batch = fixtures.synthetic_batch(2, variant, seed=seed, lanes=9 if variant != "img" else 4)
    args = harness.forward_args(batch, variant)
o64 = copy.deepcopy(oracle).double()
_, loss64, g64 = harness.train_step(o64, T._to64(args), batch["gt_wp"].double())
_, loss_ref, grads_ref = harness.train_step(oracle, args, batch["gt_wp"])
net.train()
pred = net(*T._dev_args(args))
loss = torch.nn.functional.l1_loss(pred, batch["gt_wp"].to(T.DEV), reduction="none").mean()
loss.backward()
gmax = max(t.norm().item() for t in g64.values() if t is not None)
rows = []
for name, p in net.named_parameters():
    t = g64[name]
    if t is None: continue
    n = t.norm().item()
    e_gpu = (p.grad.detach().cpu().double() - t).norm().item()
    e_cpu = (grads_ref[name].double() - t).norm().item()
    if n > 1e-6 * gmax:
        rows.append((e_gpu / max(e_cpu, 1e-12 * gmax), name, e_gpu / n, e_cpu / n))
rows.sort()
r = [x[0] for x in rows]
print("n=%d median=%.3f p90=%.3f p95=%.3f p99=%.3f max=%.3f" % (len(r), r[len(r)//2], r[int(len(r)*.9)], r[int(len(r)*.95)], r[int(len(r)*.99)], r[-1]))
by = {}
for ratio, name, eg, ec in rows:
    key = "attn" if ".attn." in name else ("ln" if ".ln" in name else ("mlp" if ".mlp." in name else ("bn" if "bn" in name else "other")))
    by.setdefault(key, []).append(ratio)
for k, v in by.items():
    v.sort(); print("  %-6s n=%4d median=%.2f p95=%.2f" % (k, len(v), v[len(v)//2], v[int(len(v)*.95)]))
for row in rows[-6:]:
    print("  %.2f %s gpu_rel=%.2e cpu_rel=%.2e" % row)
