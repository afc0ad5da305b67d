cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "winograd_input_transform or recomputes or batchnorm" 2>&1 | tail -15
 timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "winograd" 2>&1 | tail -8
 timeout 1200 python -m pytest tests/test_e2e_gpu.py -x -q -k "changes_no_bit or train_step_matches_oracle or eval_forward or lane_graph or segmented" 2>&1 | tail -15) > gpurun_out/r1_tests.log 2>&1
timeout 600 bash tools/ab_bench.sh 3 "lazy:MMFN_LAZY_BN=1" "eager:MMFN_LAZY_BN=0" > gpurun_out/r1_ab.log 2>&1
(timeout 900 python tools/grad_cosine.py --init reference --cache gpurun_out/gradref 2>&1 | tail -40
 MMFN_WINOGRAD_MIN_C=9999 timeout 600 python tools/grad_cosine.py --init reference --cache gpurun_out/gradref 2>&1 | tail -40
 timeout 900 python tools/grad_cosine.py --init closed --cache gpurun_out/gradref 2>&1 | tail -40
 MMFN_WINOGRAD_MIN_C=9999 timeout 600 python tools/grad_cosine.py --init closed --cache gpurun_out/gradref 2>&1 | tail -40) > gpurun_out/r1_cos.log 2>&1
cat gpurun_out/r1_tests.log gpurun_out/r1_ab.log gpurun_out/r1_cos.log
