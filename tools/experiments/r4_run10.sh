cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "layernorm_folded or epilogue or gemm_forms or nan_prop" 2>&1 | tail -12
 timeout 1500 python -m pytest tests/test_e2e_gpu.py -x -q 2>&1 | tail -12
 timeout 900 python -m pytest tests/test_parallel_gpu.py -x -q -k "bench_launches" 2>&1 | tail -4
 timeout 900 python -m pytest tests/test_frames_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -4
 timeout 1200 python -m pytest tests/test_parity_benchsize_gpu.py -x -q -s -k "benched_initialisation or vec_batch32_16384 or rad_batch16" 2>&1 | grep "cosine to\|loss HIP\|passed\|failed\|Error\|ratio" | cut -c1-900) > gpurun_out/r10_tests.log 2>&1
timeout 600 bash tools/ab_bench.sh 3 "fold:MMFN_LN_FOLD=1" "nofold:MMFN_LN_FOLD=0" > gpurun_out/r10_ab.log 2>&1
timeout 300 python tools/latency_bench.py > gpurun_out/r10_lat.log 2>&1
cat gpurun_out/r10_tests.log gpurun_out/r10_ab.log; tail -5 gpurun_out/r10_lat.log
