"""Gradient yardsticks shared by tests/test_parity_benchsize_gpu.py and tools/grad_cosine.py (TEST INFRASTRUCTURE: nothing
under mmfn_amd/ imports this).

The backward of phase2_train_net.py:104-108 (`loss.backward()` through 85 train-mode BatchNorms) amplifies fp32 rounding, so a
HIP gradient is judged against an fp64 evaluation of the oracle's graph, with the fp32 oracle's own distance to fp64 as the
yardstick, per backward stage (mmfn_amd.params.FlatLayout.stage_of: 0 = fusion scale 4 ... 3 = layer1 + stems + VectorNet)."""
import copy

import torch

from . import harness


def to64(a):
    if torch.is_tensor(a):
        return a.double() if a.is_floating_point() else a
    if isinstance(a, (list, tuple)):
        return type(a)(to64(x) for x in a)
    return a


def stage_of(name):
    """Same rule as mmfn_amd.params.FlatLayout.stage_of (restated here so the checker does not import the product)."""
    for s, (gpt, layer) in enumerate((("transformer4", "layer4"), ("transformer3", "layer3"), ("transformer2", "layer2"))):
        if gpt in name or layer in name:
            return s
    if name.startswith(("join.", "decoder.", "output.")) or "radar_encoder" in name:
        return 0
    return 3


GROUPS = ("head", "gpt", "vec", "img", "lid", "map", "other")


def group_of(name):
    """(stage, readiness group) of a parameter - the rule of mmfn_amd.params.FlatLayout.group_of restated: the trunk / transformer /
    VectorNet a tensor belongs to inside its backward stage.  Finer than the stage: a fault in one trunk's backward moves the
    direction of that trunk's group, not of a stage dominated by VectorNet's 16 M-parameter generator."""
    if name.startswith(("join.", "decoder.", "output.")) or "radar_encoder" in name:
        return (stage_of(name), "head")
    for key, g in (("transformer", "gpt"), ("vectornet_encoder", "vec"), ("image_encoder", "img"), ("lidar_encoder", "lid"),
                   ("img_map_encoder", "map")):
        if key in name:
            return (stage_of(name), g)
    return (stage_of(name), "other")


def oracle_gradients(oracle, args, gt_wp):
    """-> (loss32, grads32, loss64, grads64, pred32) of one train-mode step of `oracle` (weights untouched: the AdamW step of
    harness.train_step is applied to copies)."""
    o64 = copy.deepcopy(oracle).double()
    _, loss64, g64 = harness.train_step(o64, to64(args), gt_wp.double())
    del o64
    o32 = copy.deepcopy(oracle)
    pred32, loss32, g32 = harness.train_step(o32, args, gt_wp)
    return loss32, g32, loss64, g64, pred32


def stage_cosines(hip_grads, g32, g64, group=stage_of):
    """{stage: (cos(HIP, fp64), cos(CPU fp32, fp64))} over all tensors of a stage taken together."""
    acc = {}
    for name, t in g64.items():
        if t is None:
            continue
        st = group(name)
        a, b, c = hip_grads[name].detach().cpu().double().flatten(), t.flatten(), g32[name].double().flatten()
        d = acc.setdefault(st, [0.0] * 5)
        d[0] += float(torch.dot(a, b)); d[1] += float(torch.dot(a, a)); d[2] += float(torch.dot(b, b))
        d[3] += float(torch.dot(c, b)); d[4] += float(torch.dot(c, c))
    return {st: (d[0] / max((d[1] * d[2]) ** 0.5, 1e-300), d[3] / max((d[4] * d[2]) ** 0.5, 1e-300)) for st, d in sorted(acc.items())}


def tensor_rows(hip_grads, g32, g64):
    """[(stage, name, |f64|, rel err HIP, rel err CPU fp32, cos HIP, cos CPU fp32)] for every non-zero gradient tensor."""
    rows = []
    for name, t in g64.items():
        if t is None:
            continue
        b = t.flatten()
        n = b.norm().item()
        if n == 0.0:
            continue
        a, c = hip_grads[name].detach().cpu().double().flatten(), g32[name].double().flatten()
        cos = lambda u: float(torch.dot(u, b) / max(u.norm().item() * n, 1e-300))
        rows.append((stage_of(name), name, n, (a - b).norm().item() / n, (c - b).norm().item() / n, cos(a), cos(c)))
    return rows


def stage_bar(cos_cpu, factor=4.0):
    """The bar a HIP cosine has to clear, given the fp32 oracle's cosine on the same group of tensors: at most `factor` times the
    oracle's angle^2 away from the fp64 direction (1 - cos ~ angle^2 / 2), and never below 0.999 where the oracle itself reaches
    0.9999.  Why 4 and not the 2 the round-3 review proposed: measured at the benched initialisation (vec, batch 32,
    tools/grad_cosine.py) the HIP path sits at 1.6-2.0 x the oracle's 1 - cos with EVERY convolution as a
    direct implicit GEMM (longer sequential fp32 accumulation chains in the MFMA k-loop than oneDNN's blocked sums) and at
    1.8-2.3 x per stage / up to 2.8 x per trunk with the F(4x4,3x3) Winograd convolutions over the points 0, +-3/4, +-3/2
    (4.0-5.0 x over Lavin's points 0, +-1, +-2 of rounds 1-3); a wrong tap, scale or mask in ONE layer's backward lands at
    ~10000 x (tests/test_parity_benchsize_gpu.py injects one).

    The absolute allowance (1e-5 of 1 - cos, round 6; 1e-6 before): at this initialisation every path - the fp32 oracle included - sits
    0.3-0.6 % (relative) off the fp64 gradient, because 85 train-mode BatchNorms amplify fp32 rounding, and WHICH 0.5 % a path draws is
    decided by rounding-level details of its forward.  Measured when the narrow transformers' forward became the fused kernels (their
    tensors agree with the separate kernels' to 1e-7 and are exactly as far from fp64: tools/experiments/gpt_block_accuracy.py):
    per-group 1 - cos moved by up to 2.5e-5 in BOTH directions ((1, img) 3.8e-5 -> 3.3e-5, (3, gpt) 2.2e-5 -> 4.5e-5, (0, map) 5e-6 ->
    1.3e-5) with the backward untouched (MMFN_GPT_FUSED=fwd vs 0).  The (0, map) group is where the oracle's own draw is unusually
    lucky (1.5e-6, its sibling trunks 7e-6), so a purely multiplicative bar there measures the oracle's luck, not the HIP path."""
    bar = 1.0 - factor * (1.0 - cos_cpu) - 1e-5
    if cos_cpu >= 0.9999:
        bar = max(bar, 0.999)
    return bar
