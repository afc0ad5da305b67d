#!/bin/bash
# Collect the rocprofv3 evidence for a round (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of the default bench command,
#   2. PMC passes (separate runs, no tracing domains besides kernel-trace): MFMA busy / wave cycles,
#      FETCH_SIZE, WRITE_SIZE for the dominant GEMM kernels.
# Usage: bash tools/profile_round.sh r01
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# EXTRA: further bench.py flags (e.g. EXTRA="--dtype bf16" for the bf16 mode: the GEMM family is then gemm16_*)
SHORT="python $R/bench.py --no-graph --single-stream --no-cpu-baseline --no-oracle-check --no-also --steps 1 --warmup 0 --profile-steps 1 $EXTRA"
BENCH="python $R/bench.py --no-graph --single-stream --no-cpu-baseline --no-oracle-check --no-also --steps 5 --warmup 2 --profile-steps 1 $EXTRA"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT -o pmc_mfma -- $SHORT > $OUT/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o pmc_fetch -- $SHORT > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o pmc_write -- $SHORT > $OUT/pmc_write.log 2>&1
ls -la $OUT
python $R/tools/summarize_profile.py $OUT $TAG
