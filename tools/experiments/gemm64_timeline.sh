#!/bin/bash
# The instruction-level picture of the 64x64 fp32 GEMM population, two ways, both on the isolated shapes:
#   1. tools/experiments/gemm64_timeline.py: s_memtime stamps inside an experiment build of the kernel (hipcc -DMMFN_GEMM_TIMELINE
#      on the tracked source) -> per-k-tile issue timeline, prologue / stores, co-residency (no thread-trace decoder in the image);
#   2. rocprofv3 --pmc passes with the SQ instruction-mix / LDS / VMEM counters on the product library.
# Run through gpurun: bash tools/experiments/gemm64_timeline.sh [out-tag]
export TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/gemm64_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$R/tools/experiments/build
if [ ! -f $B/libmmfn_hip_tl.so ]; then
  mkdir -p $B
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DMMFN_GEMM_TIMELINE -c $R/mmfn_amd/csrc/gemm_f32.hip -o $B/gemm_f32_tl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmmfn_hip_tl.so $B/gemm_f32_tl.o $(ls $R/mmfn_amd/lib/*.o | grep -v gemm_f32.o)
fi
MMFN_HIP_LIB=$B/libmmfn_hip_tl.so MMFN_TUNING_TABLE=0 timeout 300 python $R/tools/experiments/gemm64_timeline.py > $OUT/timeline.txt 2> $OUT/timeline.err
tail -5 $OUT/timeline.err
# ---- PMC: instruction mix and LDS / VMEM cycles of the same launches (product library)
cat > $OUT/run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mmfn_amd import ops
DEV = "cuda:0"
for (T, C) in ((512, 256), (2048, 128), (8192, 64)):
    V = torch.randn(36, T, C, device=DEV); U = torch.randn(36, C, C, device=DEV); M = torch.empty(36, T, C, device=DEV)
    for tile in (2, 7):
        for _ in range(3):
            ops.gemm(V, U, M, T, C, C, C, C, C, ops.A_ROWMAJOR, ops.B_NK, batch=36, strideA=T * C, strideB=C * C, strideC=T * C, tile=tile, splitk=1)
            ops.gemm(V, U, M, T, C, C, C, C, C, ops.A_ROWMAJOR, ops.B_KN, batch=36, strideA=T * C, strideB=C * C, strideC=T * C, tile=tile, splitk=1)
x = torch.randn(6144, 512, device=DEV); w = torch.randn(512, 512, device=DEV); y = torch.empty(6144, 512, device=DEV)
for tile in (1, 3, 5):
    for _ in range(3):
        ops.linear_fwd(x, w, out=y, tile=tile, splitk=1)
torch.cuda.synchronize()
PY
S1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
S2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
S3="SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
S4="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
i=0
for S in "$S1" "$S2" "$S3" "$S4"; do
  i=$((i+1))
  MMFN_TUNING_TABLE=0 timeout 200 rocprofv3 --kernel-trace --pmc $S --output-format csv -d $OUT -o p$i -- python $OUT/run.py > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python3 - > $OUT/pmc.txt <<'PY'
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/gemm64_' + os.environ.get('TAG', 'r05')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + '/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if 'gemm_f32' in k:
            agg[(k, r['Grid_Size'], r['Workgroup_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    n = {c: sum(x) / len(x) for c, x in v.items()}
    waves = max(n.get('SQ_WAVES', 1), 1)
    mf = max(n.get('SQ_INSTS_MFMA', 0), 1)
    wc = max(n.get('SQ_WAVE_CYCLES', 1), 1)
    print("%s grid %s wg %s" % k)
    print("   per wave: MFMA %.0f VALU(non-MFMA) %.0f SALU %.0f SMEM %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f" % (
        mf / waves, (n.get('SQ_INSTS_VALU', 0) - mf) / waves, n.get('SQ_INSTS_SALU', 0) / waves, n.get('SQ_INSTS_SMEM', 0) / waves,
        n.get('SQ_INSTS_LDS', 0) / waves, n.get('SQ_INSTS_VMEM_RD', 0) / waves, n.get('SQ_INSTS_VMEM_WR', 0) / waves))
    print("   of the wave cycles: MFMA busy %.3f | waiting for any instruction to issue %.3f | waiting (any) %.3f | issuing %.3f" % (
        n.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * wc), n.get('SQ_WAIT_INST_ANY', 0) / wc, n.get('SQ_WAIT_ANY', 0) / wc, n.get('SQ_ACTIVE_INST_ANY', 0) / wc))
    print("   LDS: wait_inst_lds %.3f of wave cycles, active_inst_lds %.3f, bank-conflict cycles / idx-active cycles %.3f, addr conflicts %.0f | VMEM: inst cycles %.3f of wave cycles, active_inst_vmem %.3f" % (
        n.get('SQ_WAIT_INST_LDS', 0) / wc, n.get('SQ_ACTIVE_INST_LDS', 0) / wc, n.get('SQ_LDS_BANK_CONFLICT', 0) / max(n.get('SQ_LDS_IDX_ACTIVE', 1), 1),
        n.get('SQ_LDS_ADDR_CONFLICT', 0), n.get('SQ_INST_CYCLES_VMEM', 0) / wc, n.get('SQ_ACTIVE_INST_VMEM', 0) / wc))
    print("   raw: " + " ".join("%s=%.3g" % (c, x) for c, x in sorted(n.items())))
PY
head -80 $OUT/pmc.txt
