#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/tune16.sh > gpurun_out/tune16.log 2>&1
tail -3 gpurun_out/tune16.log
b() { python bench.py --config bf16 --no-cpu-baseline --no-oracle-check 2>/dev/null | cut -c60-100; }
for rep in 1 2; do
  echo "old table $(b)"
  echo "new table $(MMFN_TUNING_FILE16=$GRAFT_REPO_ROOT/gpurun_out/gfx950_bf16.json b)"
done
python - <<'PY'
import json
a = json.load(open("mmfn_amd/tuning/gfx950_bf16.json")); b = json.load(open("gpurun_out/gfx950_bf16.json"))
ch = [k for k in b if a.get(k) != b[k]]
print(len(a), len(b), "changed", len(ch))
PY
