#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p1 -- python $R/tools/attn_bench.py > $OUT/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_IFETCH_LEVEL SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p2 -- python $R/tools/attn_bench.py > $OUT/p2.log 2>&1
python - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_attn'
for f in sorted(glob.glob(out+'/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in rows:
        k=r['Kernel_Name'].replace('(anonymous namespace)::','')[:45]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    print(f)
    for k,v in agg.items():
        if 'attn' not in k: continue
        n=max(cnt[(k,c)] for c in v)
        print(k, 'n=%d'%n, {c: round(x/n) for c,x in v.items()})
PY
tail -3 $OUT/p1.log
