#!/bin/bash
# kernel trace of the batch-1 DrivingSession tick: which kernels make up the ~5 ms
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tick; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/tools/latency_bench.py > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, os, collections
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/tick'
rows = list(csv.DictReader(open(out + '/t_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.1f ms over %d launches" % (tot / 1e6, sum(int(r['Calls']) for r in rows)))
for r in rows[:28]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
    print("%-62s %7s %10.1f us avg %7.2f us %5.1f%%" % (n, r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
