cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "layernorm_inside" 2>&1 | grep -v "^$" | tail -30 | cut -c1-400
 timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "layernorm_folded" 2>&1 | tail -3) > gpurun_out/r13_tests.log
timeout 900 bash tools/ab_bench.sh 3 "fold:MMFN_LN_FOLD=1" "evalonly:MMFN_LN_FOLD=eval" > gpurun_out/r13_ab.log 2>&1
(MMFN_LN_FOLD=eval timeout 300 python tools/latency_bench.py 2>/dev/null | tail -1 | cut -c1-200; MMFN_LN_FOLD=0 timeout 300 python tools/latency_bench.py 2>/dev/null | tail -1 | cut -c1-200) > gpurun_out/r13_lat.log
# bf16: tune the shapes the committed table does not hold (fp32-output proj / mlp.2), then bench with both tables
timeout 900 bash tools/tune16.sh > gpurun_out/r13_tune16.log 2>&1
(timeout 400 python bench.py --config bf16 --no-cpu-baseline --no-also --no-oracle-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 committed table', d['value'], d['ms_per_step'])"
 MMFN_TUNING_FILE16=$GRAFT_REPO_ROOT/gpurun_out/gfx950_bf16.json timeout 400 python bench.py --config bf16 --no-cpu-baseline --no-also --no-oracle-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 extended table', d['value'], d['ms_per_step'])") > gpurun_out/r13_bf16.log 2>&1
cat gpurun_out/r13_tests.log gpurun_out/r13_ab.log gpurun_out/r13_lat.log; tail -3 gpurun_out/r13_tune16.log; cat gpurun_out/r13_bf16.log
