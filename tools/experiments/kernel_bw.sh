#!/bin/bash
# Per kernel of the eager single-stream step: launches, average duration, HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate
# PMC passes) and the resulting TB/s - which streaming kernels sit well below the HBM roofline.  EXTRA: further bench.py flags.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SHORT="python $R/bench.py --no-graph --single-stream --no-cpu-baseline --no-oracle-check --no-also --steps 1 --warmup 0 --profile-steps 1 $EXTRA"
rm -rf /tmp/kb_t /tmp/kb_f /tmp/kb_w
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb_t -o t -- $SHORT > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/kb_f -o p -- $SHORT > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/kb_w -o p -- $SHORT > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
dur = {}
for r in csv.DictReader(open(glob.glob("/tmp/kb_t/**/*kernel_stats.csv", recursive=True)[0])):
    dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
def pmc(d, name):
    agg = collections.Counter()
    for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
        if r["Counter_Name"] == name: agg[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    return agg
f, w = pmc("/tmp/kb_f", "FETCH_SIZE"), pmc("/tmp/kb_w", "WRITE_SIZE")
rows = []
for k, (n, us) in dur.items():
    b = (2 * f.get(k, 0) + w.get(k, 0)) * 1024 / max(n, 1)
    rows.append((n * us, k, n, us, b))
print("%-66s %6s %9s %10s %8s" % ("kernel (4 steps)", "calls", "avg us", "MB/launch", "TB/s"))
for tot, k, n, us, b in sorted(rows, reverse=True)[:45]:
    print("%-66s %6d %9.1f %10.1f %8.2f" % (k, n, us, b / 1e6, b / us / 1e6))
PY
