#!/bin/bash
# Does a later graph replay still die in hip::Graph::UpdateStreams after the tests of tests/test_frames_gpu.py?
# (round 3: yes while captured graphs were destroyed whenever Python's GC reached them; no with MMFN_KEEP_GRAPHS=1; the fix is
# graphs.Graph / drain_graveyard - destruction only with the device idle.)
mkdir -p gpurun_out/d3
F=tests/test_frames_gpu.py
T=tests/test_trainer_gpu.py
run() {
  name=$1; shift
  "$@" > gpurun_out/d3/$name.log 2>&1
  echo "$name: rc=$? segv=$(grep -c 'Segmentation' gpurun_out/d3/$name.log) $(grep -E 'passed|failed' gpurun_out/d3/$name.log | tail -1 | cut -c1-80)"
}
run all python -m pytest $F $T -m gpu -q -x
run nofused_then_trainer python -m pytest $F $T -m gpu -q -x -k "not fused_step"
run all_keep env MMFN_KEEP_GRAPHS=1 python -m pytest $F $T -m gpu -q -x
