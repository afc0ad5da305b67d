#!/bin/bash
# Kernel trace of the hipGraph-replayed step (3 streams, warm L2), grouped by (kernel, grid) so shapes can be told apart.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/graph_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o g -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also --profile-steps 1 $EXTRA > $OUT/g.log 2>&1
python3 - <<'PY'
import csv, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/graph_trace'
rows = list(csv.DictReader(open(out + '/g_kernel_trace.csv')))
agg = collections.defaultdict(lambda: [0, 0.0])
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    key = (k, r['Grid_Size_X'] + 'x' + r['Grid_Size_Y'] + 'x' + r['Grid_Size_Z'])
    a = agg[key]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(a[1] for a in agg.values())
print("kernels=%d  sum_of_durations=%.1f ms  wall=%.1f ms" % (len(rows), tot / 1e3, (t1 - t0) / 1e6))
with open(out + '/grouped.txt', 'w') as f:
    for (k, g), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write("%-52s grid=%-16s n=%5d total=%10.1f us avg=%8.2f us\n" % (k[:52], g, n, us, us / n))
PY
head -70 $OUT/grouped.txt
