#!/bin/bash
# L2 side of the isolated fp32 Winograd-domain batched GEMMs: TCC hits / misses / requests per launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/pmc_l2; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mmfn_amd import ops
from mmfn_amd.ops import A_ROWMAJOR, B_NK
DEV = "cuda:0"
for (T, C) in ((8192, 64), (2048, 128), (512, 256), (128, 512)):
    V = torch.randn(36, T, C, device=DEV); U = torch.randn(36, C, C, device=DEV); M = torch.empty(36, T, C, device=DEV)
    junk = torch.empty(64 << 20, device=DEV)
    for _ in range(3):
        junk.fill_(1.0)   # evict the operands from L2 / MALL between launches, as in the step
        ops.gemm(V, U, M, T, C, C, C, C, C, A_ROWMAJOR, B_NK, batch=36, strideA=T * C, strideB=C * C, strideC=T * C)
torch.cuda.synchronize()
PY
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z_]*\(sum\)\?" | sort -u | head -60 > $OUT/avail.txt
grep -E "^TCC_(HIT|MISS|REQ|READ|EA_RDREQ)" $OUT/avail.txt | head -20
for S in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $S --output-format csv -d $OUT -o p -- python $OUT/run.py > $OUT/p.log 2>&1
  python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_l2/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_f32' in r['Kernel_Name']:
            agg[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
  rm -rf $OUT/p* 2>/dev/null; find $OUT -name "*counter_collection.csv" -delete
done
