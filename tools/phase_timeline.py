"""Phase account of ONE hipGraph-replayed training step from a rocprofv3 kernel trace (tools/graph_timeline.sh writes the csv):
the step is cut at the kernels that only occur at the seams of the plan (engine.Engine.forward / backward_scale) -
tokens_fwd (a fusion transformer's forward starts), upsample_add_fwd (the trunks resume), gap_sum_bwd (the backward starts),
tokens_bwd (a transformer's backward ends), upsample_adj (tail of a trunk lane) - and every phase is listed with its wall time,
its kernels, the summed kernel time, the share of its wall time with a GEMM-family kernel resident and with nothing resident.

    python tools/phase_timeline.py gpurun_out/timeline/g_kernel_trace.csv

Caveat: under `rocprofv3 --kernel-trace` concurrent kernels are largely serialised (a phase's wall time approaches the sum of its
kernel times), so launch-bound phases - the C = 64 / 128 transformers with their side work - read longer than they run untraced.
Use it to rank phases and count kernels, not as a clock.
"""
import csv
import sys


def short(name):
    k = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if k.startswith('_ZN12_GLOBAL__N_1'):
        rest = k[len('_ZN12_GLOBAL__N_1'):]
        n = 0
        while n < len(rest) and rest[n].isdigit():
            n += 1
        if n:
            k = rest[n:n + int(rest[:n])]
    return k.split('<')[0]


def main(path):
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in csv.DictReader(open(path)))
    ad = [e for e in ev if e[2].startswith('adamw')]
    if len(ad) < 6:
        print("not enough steps in the trace", len(ad))
        return
    w0, w1 = ad[3][1], ad[4][1]
    win = [e for e in ev if e[0] >= w0 and e[1] <= w1]

    def starts(prefix):
        return [e for e in win if e[2].startswith(prefix)]

    tok, up, tb = starts('tokens_fwd'), starts('upsample_add_fwd'), starts('tokens_bwd_kernel')
    gapb, adj = starts('gap_sum_bwd'), starts('upsample_adj')
    if len(tok) != 4 or len(tb) != 4 or not gapb:
        print("unexpected seam kernels: tokens_fwd %d tokens_bwd %d gap_sum_bwd %d" % (len(tok), len(tb), len(gapb)))
        return
    cuts = [("fwd ingest + stems + layer1 (3 lanes)", w0)]
    for k in range(4):
        cuts.append(("fwd transformer %d" % (k + 1), tok[k][0]))
        nxt = [u for u in up if u[0] > tok[k][0]]
        cuts.append(("fwd layer%d (3 lanes)" % (k + 2) if k < 3 else "fwd fuse + head + loss", nxt[0][0]))
    cuts.append(("bwd head", [e for e in win if e[2].startswith('gap_sum_fwd')][0][1]))
    t = gapb[0][1]
    for k in range(4):
        cuts.append(("bwd transformer %d" % (4 - k), t))
        t = tb[k][1]
        cuts.append(("bwd layer%d (3 lanes)" % (4 - k) if k < 3 else "bwd layer1 + stems + VectorNet (3 lanes)", t))
        if k < 3:
            nxt_tb = tb[k + 1][0]
            tails = [a for a in adj if t < a[1] < nxt_tb]
            # the lanes' last launches are their upsample_adj; the next transformer's backward starts after the join
            t = max(a[1] for a in tails[:3]) if tails else t
    cuts.append(("AdamW", ad[4][0]))
    cuts.append(("end", w1))
    print("# phases of one replayed step (%d kernels, wall %.2f ms)" % (len(win), (w1 - w0) / 1e6))
    print("%-44s %8s %7s %10s %7s %7s" % ("phase", "wall ms", "kernels", "kernel ms", "gemm %", "idle %"))
    for (name, a), (_, b) in zip(cuts[:-1], cuts[1:]):
        if b <= a:
            continue
        inside = [e for e in win if e[0] < b and e[1] > a]
        pts = []
        for s, e, k in inside:
            g = k.startswith('gemm') or k.startswith('splitk')
            pts.append((max(s, a), 1, g)); pts.append((min(e, b), -1, g))
        pts.sort()
        act = gem = 0
        last = a
        t_g = t_idle = 0
        for tt, d, g in pts:
            dt = tt - last
            if dt > 0:
                if gem > 0:
                    t_g += dt
                elif act == 0:
                    t_idle += dt
            act += d
            gem += d if g else 0
            last = tt
        t_idle += max(0, b - last)
        ksum = sum(min(e, b) - max(s, a) for s, e, _ in inside)
        print("%-44s %8.3f %7d %10.3f %7.1f %7.1f" % (name, (b - a) / 1e6, sum(1 for e in inside if e[0] >= a), ksum / 1e6,
                                                      100.0 * t_g / (b - a), 100.0 * t_idle / (b - a)))


if __name__ == "__main__":
    main(sys.argv[1])
