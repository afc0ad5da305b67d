cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "winograd_input_transform or recomputes" 2>&1 | tail -5
 timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "nan_propagates or epilogue" 2>&1 | tail -5
 timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_parallel_gpu.py tests/test_trainer_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r2_tests.log 2>&1
timeout 600 bash tools/ab_bench.sh 3 "lazy:MMFN_LAZY_BN=1" "eager:MMFN_LAZY_BN=0" > gpurun_out/r2_ab.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 1 0; do
  O=$R/gpurun_out/ks_lazy$m; rm -rf $O; mkdir -p $O
  MMFN_LAZY_BN=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python $R/bench.py --no-graph --single-stream --no-cpu-baseline --no-oracle-check --no-also --steps 5 --warmup 2 --profile-steps 1 > $O/trace.log 2>&1
  python $R/tools/summarize_profile.py $O lazy$m > /dev/null 2>&1
  cp $O/summary/lazy${m}_kernel_stats.txt $R/gpurun_out/ 2>/dev/null
  rm -rf $O
done
cd $R
(MMFN_WINOGRAD_MIN_C=128 timeout 600 python tools/grad_cosine.py --init reference --cache gpurun_out/gradref --tensors 2 2>&1 | grep "stage\|loss\|\[" | grep -v "    stage"
 MMFN_WINOGRAD_MIN_C=256 timeout 600 python tools/grad_cosine.py --init reference --cache gpurun_out/gradref --tensors 2 2>&1 | grep "stage\|loss\|\[" | grep -v "    stage") > gpurun_out/r2_cos.log 2>&1
rm -f gpurun_out/gradref*
timeout 300 bash tools/ab_bench.sh 1 "c64:MMFN_WINOGRAD_MIN_C=64" "c128:MMFN_WINOGRAD_MIN_C=128" "c256:MMFN_WINOGRAD_MIN_C=256" > gpurun_out/r2_ab2.log 2>&1
cat gpurun_out/r2_tests.log gpurun_out/r2_ab.log gpurun_out/r2_cos.log gpurun_out/r2_ab2.log
head -45 gpurun_out/lazy1_kernel_stats.txt; head -45 gpurun_out/lazy0_kernel_stats.txt
