#!/bin/bash
# HBM traffic counters for the GEMM microbench (separate passes, counters only + kernel-trace).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
export ITERS=3 CASES=${CASES:-lin} ONLY=${ONLY:-fwd}
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o $C -- python $R/tools/gemm_bench.py > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
ls $OUT
python - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_traffic'
for f in sorted(glob.glob(out+'/*counter_collection.csv')):
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'].replace('(anonymous namespace)::','')[:50], r['Grid_Size'])
        agg[k]+=float(r['Counter_Value']); cnt[k]+=1
    print(f)
    for k,v in agg.items():
        if 'gemm' in k[0]: print(k, 'n=%d'%cnt[k], 'per launch: %.1f' % (v/cnt[k]))
PY
