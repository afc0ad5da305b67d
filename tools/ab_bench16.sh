#!/bin/bash
# tools/ab_bench.sh for the bf16 mode
ROUNDS=$1; shift
R=${GRAFT_REPO_ROOT:-.}
OUT=$R/gpurun_out/ab16; mkdir -p $OUT; rm -f $OUT/*.txt
for r in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    ( IFS=';'; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
      python $R/bench.py --dtype bf16 --no-oracle-check --no-cpu-baseline --no-also --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" >> $OUT/$name.txt )
  done
done
for spec in "$@"; do
  name=${spec%%:*}
  python - "$name" "$OUT/$name.txt" <<'PY'
import sys
v = sorted(float(x) for x in open(sys.argv[2]).read().split())
med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
print("%-16s median %.3f ms  (%.1f samples/s)   runs: %s" % (sys.argv[1], med, 32e3 / med, " ".join("%.2f" % x for x in v)))
PY
done
