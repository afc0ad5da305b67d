#!/bin/bash
# PMC counters of the isolated fp32 Winograd-domain batched GEMMs (36 x [tiles x Cin] . [Cin x Cout], 64x64 tiles): instruction mix per
# k-tile and wave.  Run through gpurun: bash tools/experiments/gemm32_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gemm32
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mmfn_amd import ops
from mmfn_amd.ops import A_ROWMAJOR, B_NK, B_KN
DEV = "cuda:0"
for (T, C) in ((8192, 64), (2048, 128), (512, 256), (128, 512)):
    V = torch.randn(36, T, C, device=DEV); U = torch.randn(36, C, C, device=DEV); M = torch.empty(36, T, C, device=DEV)
    for _ in range(3):
        ops.gemm(V, U, M, T, C, C, C, C, C, A_ROWMAJOR, B_NK, batch=36, strideA=T * C, strideB=C * C, strideC=T * C)
        ops.gemm(V, U, M, T, C, C, C, C, C, A_ROWMAJOR, B_KN, batch=36, strideA=T * C, strideB=C * C, strideC=T * C)
torch.cuda.synchronize()
PY
S1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
S2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
i=0
for S in "$S1" "$S2"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $S --output-format csv -d $OUT -o p$i -- python $OUT/run.py > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/pmc_gemm32'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + '/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if 'gemm_f32' in k:
            agg[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    n = {c: sum(x) / len(x) for c, x in v.items()}
    waves = n.get('SQ_WAVES', 1)
    mf = n.get('SQ_INSTS_MFMA', 0)
    print(k)
    print("   per wave: MFMA %.0f VALU %.0f SALU %.0f LDS %.0f VMEM %.0f | per 16 MFMAs (one 32-deep k-tile): VALU %.1f SALU %.1f LDS %.1f VMEM %.1f | wave quad-cycles/MFMA %.1f  mfma busy %.2f" % (
        mf / waves, n.get('SQ_INSTS_VALU', 0) / waves, n.get('SQ_INSTS_SALU', 0) / waves, n.get('SQ_INSTS_LDS', 0) / waves, n.get('SQ_INSTS_VMEM_RD', 0) / waves,
        16 * n.get('SQ_INSTS_VALU', 0) / max(mf, 1), 16 * n.get('SQ_INSTS_SALU', 0) / max(mf, 1), 16 * n.get('SQ_INSTS_LDS', 0) / max(mf, 1), 16 * n.get('SQ_INSTS_VMEM_RD', 0) / max(mf, 1),
        n.get('SQ_WAVE_CYCLES', 0) / max(mf, 1), n.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(4 * n.get('SQ_WAVE_CYCLES', 1), 1)))
PY
