#!/bin/bash
# Dev tool: PMC counters of the fused GPT-block kernels (tools/gpt_block_bench.py) - run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gpt
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p1 -- python $R/tools/gpt_block_bench.py > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p2 -- python $R/tools/gpt_block_bench.py > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p3 -- python $R/tools/gpt_block_bench.py > $OUT/p3.log 2>&1
python - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_gpt'
for f in sorted(glob.glob(out+'/**/*counter_collection.csv', recursive=True)):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in rows:
        k=r['Kernel_Name'].replace('(anonymous namespace)::','')[:48]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        cnt[(k,r['Counter_Name'])]+=1
    print(os.path.basename(f))
    for k,v in agg.items():
        if 'gpt_' not in k: continue
        n=max(cnt[(k,c)] for c in v)
        print(' ', k, 'n=%d'%n, {c: round(x/n) for c,x in v.items()})
PY
