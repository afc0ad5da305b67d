cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "recomputes or winograd_input" 2>&1 | tail -12
 timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -s -k "winograd" 2>&1 | grep -v "^$" | tail -12
 timeout 1500 python -m pytest tests/test_e2e_gpu.py -x -q -k "changes_no_bit or train_step_matches or eval_forward or golden" 2>&1 | tail -8
 timeout 1200 python -m pytest tests/test_parity_benchsize_gpu.py -x -q -s -k "benched_initialisation" 2>&1 | grep "cosine to\|loss HIP\|passed\|failed\|Error" | cut -c1-3000) > gpurun_out/r4_tests.log 2>&1
timeout 900 python tools/trainer_bench.py > gpurun_out/r4_trainer.log 2>&1
timeout 300 bash tools/ab_bench.sh 2 "new:MMFN_LAZY_BN=1" > gpurun_out/r4_ab.log 2>&1
(timeout 900 python tools/grad_cosine.py --init reference --tensors 2 2>&1 | grep "stage\|loss\|\[" | grep -v "    stage"
 timeout 900 python tools/grad_cosine.py --init closed --tensors 2 2>&1 | grep "stage\|loss\|\[" | grep -v "    stage") > gpurun_out/r4_cos.log 2>&1
cat gpurun_out/r4_tests.log gpurun_out/r4_trainer.log gpurun_out/r4_ab.log gpurun_out/r4_cos.log
