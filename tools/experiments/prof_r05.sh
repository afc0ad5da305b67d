#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
bash $R/tools/profile_round.sh r05 > $R/gpurun_out/prof_r05.log 2>&1
cp $R/gpurun_out/prof_r05/summary/* $R/gpurun_out/profiles/ 2>/dev/null
PROFILE_FAMILY=gemm16 EXTRA="--dtype bf16" bash $R/tools/profile_round.sh r05_bf16 > $R/gpurun_out/prof_r05_bf16.log 2>&1
cp $R/gpurun_out/prof_r05_bf16/summary/* $R/gpurun_out/profiles/ 2>/dev/null
rm -rf $R/gpurun_out/prof_r05 $R/gpurun_out/prof_r05_bf16
bash $R/tools/graph_timeline.sh > /dev/null 2>&1
cat $R/gpurun_out/timeline/summary.txt $R/gpurun_out/timeline/phases.txt > $R/gpurun_out/profiles/r05_graph_timeline.txt
EXTRA="--dtype bf16" bash $R/tools/graph_timeline.sh > /dev/null 2>&1
cat $R/gpurun_out/timeline/summary.txt $R/gpurun_out/timeline/phases.txt > $R/gpurun_out/profiles/r05_bf16_graph_timeline.txt
bash $R/tools/profile_configs.sh r05 > $R/gpurun_out/prof_cfg.log 2>&1
cp $R/gpurun_out/prof_cfg/r05_* $R/gpurun_out/profiles/ 2>/dev/null
ls $R/gpurun_out/profiles
python $R/tools/phase_clock.py > $R/gpurun_out/profiles/r05_phase_clock_f32.txt 2> $R/gpurun_out/clock_f32.err
python $R/tools/phase_clock.py --dtype bf16 > $R/gpurun_out/profiles/r05_phase_clock_bf16.txt 2> $R/gpurun_out/clock_bf16.err
