#!/bin/bash
mkdir -p gpurun_out/d5
P=tests/test_parallel_gpu.py
T=tests/test_trainer_gpu.py
run() {
  name=$1; shift
  "$@" > gpurun_out/d5/$name.log 2>&1
  echo "$name: rc=$? segv=$(grep -c 'Segmentation' gpurun_out/d5/$name.log) $(grep -E 'passed|failed' gpurun_out/d5/$name.log | tail -1 | cut -c1-80)"
}
run parked python -m pytest $P $T -m gpu -q -x
run parked2 python -m pytest $P $T -m gpu -q -x
run nocapture env MMFN_DBG_RAW_CAPTURE=none python -m pytest $P $T -m gpu -q -x
run bf16tests python -m pytest tests/test_bf16_mode_gpu.py -m gpu -q -x
python -m pytest tests -m gpu -x -q > gpurun_out/d5/full.log 2>&1; tail -3 gpurun_out/d5/full.log
