#!/bin/bash
# Re-measure (tile, LDS stages) of the stride-2 data-gradient shapes only (their kernel changed: parity-pure row tiles), keeping the
# rest of mmfn_amd/tuning/gfx950_bf16.json.  Usage (through gpurun): bash tools/retune16_s2dgrad.sh; result in gpurun_out/gfx950_bf16.json
R=${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}
cd $R
mkdir -p gpurun_out
python - <<'PY'
import json
t = json.load(open("mmfn_amd/tuning/gfx950_bf16.json"))
keep = {}
for k, v in t.items():
    head, conv = k.split("|")
    if head.startswith("2,") and conv and int(conv.split(",")[8]) == 2:
        continue
    keep[k] = v
json.dump(keep, open("gpurun_out/gfx950_bf16.json", "w"), indent=0)
print("kept", len(keep), "of", len(t))
PY
export MMFN_AUTOTUNE16=1 MMFN_TUNING_FILE16=$R/gpurun_out/gfx950_bf16.json
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from mmfn_amd import ops16
from mmfn_amd.config import GlobalConfig
from mmfn_amd.model import MMFN, MMFNImg
dev = torch.device("cuda", 0)
for cls, variant, B in ((MMFN, "vec", 32), (MMFNImg, "img", 32), (MMFN, "vec", 2), (MMFN, "vec", 1)):
    torch.manual_seed(42)
    net = cls(GlobalConfig(act_dtype="bf16"), dev)
    net.train()
    inp, gt = bench.synth_inputs(B, dev, seed=42, variant=variant)
    for _ in range(2):
        net.train_step(inp, gt)
    torch.cuda.synchronize()
    print(variant, B, len(ops16._tuned), "shapes", flush=True)
    del net
ops16.save_tuning(os.environ["MMFN_TUNING_FILE16"])
PY
python - <<'PY'
import json
a = json.load(open("mmfn_amd/tuning/gfx950_bf16.json")); b = json.load(open("gpurun_out/gfx950_bf16.json"))
for k in sorted(b):
    if a.get(k) != b[k]:
        print(k, a.get(k), "->", b[k])
PY
for i in 1 2; do echo "old table $(python bench.py --config bf16 --no-cpu-baseline --no-oracle-check 2>/dev/null | cut -c60-130)"; echo "new table $(MMFN_TUNING_FILE16=$R/gpurun_out/gfx950_bf16.json MMFN_AUTOTUNE16=0 python bench.py --config bf16 --no-cpu-baseline --no-oracle-check 2>/dev/null | cut -c60-130)"; done
