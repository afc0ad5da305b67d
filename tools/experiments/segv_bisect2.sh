#!/bin/bash
# Round 4: a graph replay dies (SIGSEGV inside hipGraphLaunch) some tests AFTER the tests that create / destroy RCCL communicators.
# Which ingredient?  Each line is a fresh python process over the same two files.
mkdir -p gpurun_out/d4
P=tests/test_parallel_gpu.py
T=tests/test_trainer_gpu.py
run() {
  name=$1; shift
  "$@" > gpurun_out/d4/$name.log 2>&1
  echo "$name: rc=$? segv=$(grep -c 'Segmentation' gpurun_out/d4/$name.log) $(grep -E 'passed|failed' gpurun_out/d4/$name.log | tail -1 | cut -c1-80)"
}
run asis python -m pytest $P $T -m gpu -q -x
run asis2 python -m pytest $P $T -m gpu -q -x
run comm_keep env MMFN_DBG_COMM_KEEP=1 python -m pytest $P $T -m gpu -q -x
run graphs_keep env MMFN_DBG_KEEP_GRAPHS=1 python -m pytest $P $T -m gpu -q -x
run both_keep env MMFN_DBG_KEEP_GRAPHS=1 MMFN_DBG_COMM_KEEP=1 python -m pytest $P $T -m gpu -q -x
run no_raw python -m pytest $P $T -m gpu -q -x -k "not c_abi_single_rank"
run no_bench python -m pytest $P $T -m gpu -q -x -k "not bench_launches and not two_ranks"
