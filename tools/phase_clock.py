"""Untraced clock of the replayed training step, phase by phase.

rocprofv3's kernel trace serialises concurrent kernels and stretches short ones (tools/phase_timeline.py reads 36 ms for a step
that replays in 32, and 24 for the bf16 step that replays in 17.7), so it ranks phases but is no clock.  Here the step is captured
into a hipGraph 20 times, each time with every launch AFTER one more seam of the plan suppressed (the library handle is swapped
for a proxy whose entry points return 0 without launching once the seam has been passed: forks and joins are still recorded, the
kernels are not), and each truncated graph is replayed on its own, unprofiled.  The difference between consecutive prefixes is
what the phase adds to the step with everything before it in place.

    python tools/phase_clock.py [--dtype bf16] [--batch 32] [--variant vec]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmfn_amd import _lib, ops  # noqa: E402
from mmfn_amd.config import GlobalConfig  # noqa: E402
from mmfn_amd.graphs import Recorder  # noqa: E402
from mmfn_amd.model import MMFN, MMFNImg, MMFNRad  # noqa: E402


class Proxy(object):
    """Stands in for the ctypes handle: past the cut every entry point that takes a stream (i.e. launches) returns 0 instead;
    the host-side helpers (workspace sizes, partial-row counts, struct sizes) keep answering."""

    def __init__(self, real):
        self._real = real
        self.muted = False
        self.in_fork = False    # inside Engine._branches (the three trunk lanes)
        self.lanes = None       # --lanes: the lanes whose launches are kept inside a fork (None: all)
        self.no_side = False    # --side: drop the transformers' side work (weight / bias gradients, reductions: lanes 1, 2 outside forks)

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        sig = _lib._SIGNATURES.get(name)
        import ctypes
        if sig is None or not sig[1] or sig[1][-1] is not ctypes.c_void_p:
            return fn

        def call(*a):
            if self.muted or (self.in_fork and self.lanes is not None and ops.current_lane() not in self.lanes):
                return 0
            if self.no_side and not self.in_fork and ops.current_lane() != 0:
                return 0
            return fn(*a)
        return call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--variant", default="vec", choices=["vec", "img", "rad"])
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--side", action="store_true", help="the backward of each transformer with and without its side-stream work")
    ap.add_argument("--lanes", action="store_true", help="instead of the phases: the whole step with only lane 0 / 1 / 2 / none / all "
                                                         "of the trunk forks launching - what each lane costs alone and what they cost together")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    cls = {"vec": MMFN, "img": MMFNImg, "rad": MMFNRad}[args.variant]
    net = cls(GlobalConfig(act_dtype="bf16" if args.dtype == "bf16" else "f32"), dev).train()
    inp, gt = bench.synth_inputs(args.batch, dev, seed=42, variant=args.variant)
    eng = net._engine_for()
    for _ in range(2):
        eng.train_step(inp, gt, lr=1e-4)
    torch.cuda.synchronize()
    eng.set_hyper(eng.hyper_rows(lr=1e-4))
    proxy = Proxy(_lib.lib())
    _lib._lib = proxy

    # ---- seams: (label, object, attribute, which call of it) - the cut falls AFTER that call returns
    seams = []
    state = {"n": 0, "cut": None}

    def seam(label):
        state["n"] += 1
        if state["cut"] is not None and state["n"] == state["cut"]:
            proxy.muted = True
        if state["cut"] is None:
            seams.append(label)

    def wrap(obj, attr, label_of):
        real = getattr(obj, attr)
        calls = {"k": 0}

        def f(*a, **kw):
            out = real(*a, **kw)
            label = label_of(calls["k"])
            calls["k"] += 1
            if label:
                seam(label)
            return out
        f._calls = calls
        setattr(obj, attr, f)
        return f

    fwd_lane_labels = ["fwd ingest + stems + layer1 (3 lanes)", "fwd layer2 (3 lanes)", "fwd layer3 (3 lanes)", "fwd layer4 (3 lanes)"]
    bwd_lane_labels = ["bwd layer4 (3 lanes)", "bwd layer3 (3 lanes)", "bwd layer2 (3 lanes)", "bwd layer1 + stems + map branch (3 lanes)"]
    wrapped = [wrap(eng, "_branches", lambda k: (fwd_lane_labels + bwd_lane_labels)[k] if k < 8 else None)]
    for s, g in enumerate(eng.gpts):
        wrapped.append(wrap(g, "fwd", lambda k, s=s: "fwd transformer %d" % (s + 1)))
        wrapped.append(wrap(g, "bwd", lambda k, s=s: "bwd transformer %d" % (s + 1)))
    wrapped.append(wrap(eng.head, "fwd", lambda k: "fwd fuse + head + loss"))
    wrapped.append(wrap(eng, "backward_begin", lambda k: "bwd head"))
    wrapped.append(wrap(eng, "optimizer_step", lambda k: "AdamW"))

    def body():
        state["n"] = 0
        proxy.muted = False
        for w in wrapped:
            w._calls["k"] = 0
        ops.rng_advance(eng.rng_state)
        eng.forward(inp, True, gt)
        eng.backward_and_step(None, lr=1e-4)
        proxy.muted = False

    body()   # eager: collects the seam labels in program order
    torch.cuda.synchronize()
    n_seams = len(seams)

    def clock(cut):
        state["cut"] = cut
        rec = Recorder(eng)
        rec.capture(body)
        for _ in range(3):
            rec.replay()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(args.reps):
                rec.replay()
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / args.reps * 1e3
            best = t if best is None else min(best, t)
        del rec
        return best

    if args.side:
        real_branches = wrapped[0]

        def fork(*a, **kw):
            proxy.in_fork = True
            try:
                return real_branches(*a, **kw)
            finally:
                proxy.in_fork = False
        eng._branches = fork
        prev = 0.0
        print("# what the transformers' side work costs (%s B=%d %s): each backward phase with and without the side streams' launches" % (args.variant, args.batch, args.dtype))
        n_seams = len(seams)
        for k in range(1, n_seams + 1):
            if not seams[k - 1].startswith("bwd transformer"):
                continue
            proxy.no_side = False
            t_prev = clock(k - 1)
            t_full = clock(k)
            proxy.no_side = True
            t_chain = clock(k) - (clock(k - 1) - 0.0)
            proxy.no_side = False
            # prefix without side work in EARLIER phases differs too, so compare phase additions under both settings
            print("%-22s with side work +%.3f ms   chain only +%.3f ms" % (seams[k - 1], t_full - t_prev, t_chain))
        return
    if args.lanes:
        real_branches = wrapped[0]

        def fork(*a, **kw):
            proxy.in_fork = True
            try:
                return real_branches(*a, **kw)
            finally:
                proxy.in_fork = False
        eng._branches = fork
        names = {0: "lane 0 (camera ResNet-34, main stream)", 1: "lane 1 (LiDAR ResNet-18)", 2: "lane 2 (map branch)"}
        res = {}
        for tag, keep in (("none", set()), ("0", {0}), ("1", {1}), ("2", {2}), ("all", None)):
            proxy.lanes = keep
            res[tag] = clock(None)
        base = res["none"]
        print("# trunk lanes of the replayed step alone and together (%s B=%d %s): whole-step time with only that lane's launches inside the forks" % (args.variant, args.batch, args.dtype))
        print("no lane launches (transformers, head, AdamW, fork / join only)   %8.3f ms" % base)
        for k in (0, 1, 2):
            print("%-64s %8.3f ms  (+%.3f)" % ("only " + names[k], res[str(k)], res[str(k)] - base))
        print("%-64s %8.3f ms  (+%.3f; the lanes alone add up to +%.3f, the longest is +%.3f)" % (
            "all three lanes", res["all"], res["all"] - base, sum(res[str(k)] - base for k in (0, 1, 2)), max(res[str(k)] - base for k in (0, 1, 2))))
        return

    prev = 0.0
    rows = []
    for k in range(1, n_seams + 1):
        t = clock(k)
        rows.append((seams[k - 1], t - prev, t))
        prev = t
    print("# untraced phase clock: %s B=%d %s; prefix graphs replayed alone, best of 3 x %d replays" % (args.variant, args.batch, args.dtype, args.reps))
    print("%-46s %9s %10s" % ("phase", "adds ms", "prefix ms"))
    for name, d, t in rows:
        print("%-46s %9.3f %10.3f" % (name, d, t))
    tf = sum(d for n, d, _ in rows if n.startswith("fwd transformer") or n.startswith("bwd transformer"))
    tl = sum(d for n, d, _ in rows if "lanes" in n)
    print("transformers %.2f ms, trunk lanes %.2f ms, rest %.2f ms, whole step %.2f ms" % (tf, tl, prev - tf - tl, prev))


if __name__ == "__main__":
    main()
