cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "layernorm_inside" 2>&1 | tail -3
 timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "layernorm_folded" 2>&1 | tail -3) > gpurun_out/r12_tests.log
timeout 900 bash tools/ab_bench.sh 3 "fold:MMFN_LN_FOLD=1" "nofold:MMFN_LN_FOLD=0" > gpurun_out/r12_ab.log 2>&1
(MMFN_LN_FOLD=1 timeout 300 python tools/latency_bench.py 2>/dev/null | tail -1 | cut -c1-200; MMFN_LN_FOLD=0 timeout 300 python tools/latency_bench.py 2>/dev/null | tail -1 | cut -c1-200) > gpurun_out/r12_lat.log
cd /tmp && export TMPDIR=/tmp
for m in 1; do
  O=$R/gpurun_out/ks_fold$m; rm -rf $O; mkdir -p $O
  MMFN_LN_FOLD=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python $R/bench.py --no-graph --single-stream --no-cpu-baseline --no-oracle-check --no-also --steps 5 --warmup 2 --profile-steps 1 > $O/trace.log 2>&1
  python - "$O" "$m" <<'PY'
import csv, glob, sys
out, m = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("LN_FOLD=%s total kernel time per step %.3f ms, launches per step %d" % (m, tot / 1e6 / 10, sum(int(r["Calls"]) for r in rows) // 10))
for r in rows:
    n = r["Name"]
    if "gemm_f32_fast_kernel<0, 0" in n or "layernorm_fwd" in n:
        print("   %-70s calls %5s avg %8.2f us total %9.1f us" % (n.replace("(anonymous namespace)::", "").replace("void ", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
  rm -rf $O
done > $R/gpurun_out/r12_ks.log 2>&1
cd $R; cat gpurun_out/r12_tests.log gpurun_out/r12_ab.log gpurun_out/r12_lat.log gpurun_out/r12_ks.log
