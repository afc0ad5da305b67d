#!/bin/bash
# PMC counters of ONE isolated bf16 convolution launch shape (layer4: 8x8, 512 channels; 64x64 tiles, 4 stages): what the waves of a
# one-block-per-CU grid do with their cycles.  Run through gpurun: bash tools/experiments/conv16_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_conv16
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mmfn_amd import ops16
DEV = "cuda:0"; BF = torch.bfloat16
B, H, C = int(os.environ.get("PB", 32)), int(os.environ.get("PH", 8)), int(os.environ.get("PC", 512))
x = torch.randn(B, H, H, C, device=DEV).to(BF); w = (torch.randn(C, 3, 3, C, device=DEV) * 0.05).to(BF)
y = torch.empty(B, H, H, C, dtype=BF, device=DEV)
M, N, K = B * H * H, C, 9 * C
a2 = torch.randn(M, K, device=DEV).to(BF)
for _ in range(5):
    ops16.conv2d_fwd(x, w, 1, 1, y, tile=2, stages=4)
    ops16.gemm16(ops16.G16_NT, a2, w.view(C, K), y, M, N, K, K, K, N, tile=2, stages=4)
torch.cuda.synchronize()
PY
S1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
S2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
S3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
i=0
for S in "$S1" "$S2" "$S3"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $S --output-format csv -d $OUT -o p$i -- python $OUT/run.py > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/pmc_conv16'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + '/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if 'gemm16' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k)
    for c in sorted(v):
        xs = v[c]
        print("   %-28s %14.0f  (n=%d)" % (c, sum(xs) / len(xs), len(xs)))
PY
