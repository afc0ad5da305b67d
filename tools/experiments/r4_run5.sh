cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r5_suite.log 2>&1
tail -30 gpurun_out/r5_suite.log
timeout 900 bash tools/ab_bench.sh 3 "points34:MMFN_LAZY_BN=1" "lavin:MMFN_HIP_LIB=/root/repo/mmfn_amd/lib/exp/libmmfn_hip_lavin.so" > gpurun_out/r5_ab.log 2>&1
cat gpurun_out/r5_ab.log
