cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_gpu.py -x -q -k "winograd or batchnorm or recomputes" 2>&1 | tail -5
 timeout 1500 python -m pytest tests/test_e2e_gpu.py -x -q -k "changes_no_bit or driving_session or train_step_matches" 2>&1 | tail -8
 timeout 900 python -m pytest tests/test_trainer_gpu.py -x -q -k "packed or graph_replay_epoch" 2>&1 | tail -5
 timeout 1200 python -m pytest tests/test_parity_benchsize_gpu.py -x -q -s -k "benched_initialisation" 2>&1 | grep -v "^$" | tail -25) > gpurun_out/r3_tests.log 2>&1
timeout 600 bash tools/ab_bench.sh 3 "lazy:MMFN_LAZY_BN=1" "eager:MMFN_LAZY_BN=0" > gpurun_out/r3_ab.log 2>&1
timeout 900 python tools/trainer_bench.py > gpurun_out/r3_trainer.log 2>&1
timeout 900 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ks; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python $R/bench.py --no-graph --single-stream --no-cpu-baseline --no-oracle-check --no-also --steps 5 --warmup 2 --profile-steps 1 > $O/trace.log 2>&1
python $R/tools/summarize_profile.py $O r3 > /dev/null 2>&1
cp $O/summary/r3_kernel_stats.txt $R/gpurun_out/ 2>/dev/null
rm -rf $O
cd $R
cat gpurun_out/r3_tests.log gpurun_out/r3_ab.log gpurun_out/r3_trainer.log gpurun_out/r3_bench.json
tail -3 gpurun_out/r3_bench.err
grep "wino4\|col_partial\|bn_\|calls" gpurun_out/r3_kernel_stats.txt
