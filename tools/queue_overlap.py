"""Per-queue concurrency of one replayed step from a rocprofv3 kernel trace (tools/graph_timeline.sh writes it):
how long 0/1/2/3 hardware queues had a kernel resident, per-queue busy time, pairwise overlap.  usage: queue_overlap.py trace.csv [dump_from_us dump_to_us]"""
import collections
import csv
import itertools
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k, r['Queue_Id']))
ev.sort()
ad = [e for e in ev if e[2].startswith('adamw')]
w0, w1 = ad[3][1], ad[4][1]
win = [e for e in ev if e[0] >= w0 and e[1] <= w1]
print("step wall %.2f ms, %d kernels; kernels per queue: %s" % ((w1 - w0) / 1e6, len(win), dict(collections.Counter(e[3] for e in win))))
pts = []
for s, e, k, q in win:
    pts.append((s, 1, q)); pts.append((e, -1, q))
pts.sort()
act = collections.Counter(); last = w0; hist = collections.Counter()
for t, d, q in pts:
    hist[sum(1 for v in act.values() if v > 0)] += t - last
    act[q] += d; last = t
for n in sorted(hist):
    print("queues with a resident kernel = %d : %6.2f ms (%4.1f %%)" % (n, hist[n] / 1e6, 100.0 * hist[n] / (w1 - w0)))
byq = collections.defaultdict(list)
for e in win:
    byq[e[3]].append(e)


def overlap(a, b):
    i = j = t = 0
    while i < len(a) and j < len(b):
        s = max(a[i][0], b[j][0]); e = min(a[i][1], b[j][1])
        if e > s:
            t += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return t


for q in sorted(byq):
    print("queue %s busy %.2f ms" % (q, sum(e[1] - e[0] for e in byq[q]) / 1e6))
for p, q in itertools.combinations(sorted(byq), 2):
    print("queues %s,%s overlap %.2f ms" % (p, q, overlap(byq[p], byq[q]) / 1e6))
if len(sys.argv) > 3:
    lo, hi = float(sys.argv[2]), float(sys.argv[3])
    qs = sorted(byq)
    for s, e, k, q in win:
        t = (s - w0) / 1e3
        if lo < t < hi:
            print("%8.1f %6.1f  %s q%s %s" % (t, (e - s) / 1e3, "      " * qs.index(q), q, k[:44]))
