for i in 1 2; do
for spec in "table:X=1" "big64:MMFN_G16_PREFER_BIG=1" "big32:MMFN_G16_PREFER_BIG=1;MMFN_G16_BIG_MIN_TILES=32" "big128:MMFN_G16_PREFER_BIG=1;MMFN_G16_BIG_MIN_TILES=128" "mid:MMFN_G16_PREFER_BIG=2"; do
name=${spec%%:*}; envs=${spec#*:}
( IFS=';'; for kv in $envs; do export "$kv"; done; python bench.py --dtype bf16 --no-cpu-baseline --no-oracle-check --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', r['ms_per_step'], r['value'])" )
done; done
