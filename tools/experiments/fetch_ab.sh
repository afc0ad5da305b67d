#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SHORT="python $R/bench.py --no-graph --single-stream --no-cpu-baseline --no-oracle-check --no-also --steps 1 --warmup 0 --profile-steps 1"
for v in base xcd; do
  if [ $v = xcd ]; then export MMFN_HIP_LIB=$R/mmfn_amd/lib/exp/libmmfn_hip_xcd.so; fi
  rm -rf /tmp/pf_$v
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$v -o p -- $SHORT > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
f = glob.glob("/tmp/pf_%s/**/*counter_collection.csv" % v, recursive=True)[0]
agg = collections.Counter(); n = collections.Counter(); tot = 0.0
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    agg[k] += float(r["Counter_Value"]); n[k] += 1; tot += float(r["Counter_Value"])
print(v, "FETCH_SIZE all kernels, raw KiB -> GB per step (x2 corrected, 4 steps): %.2f" % (tot * 2 * 1024 / 4 / 1e9))
for k, val in agg.most_common(6):
    print("   %-62s n=%5d  per launch %8.1f MB (x2)" % (k, n[k], val * 2 * 1024 / n[k] / 1e6))
PY
done
