#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
bash $R/tools/profile_round.sh r04b > $R/gpurun_out/prof_r04b.log 2>&1
cp $R/gpurun_out/prof_r04b/summary/* $R/gpurun_out/profiles/ 2>/dev/null
PROFILE_FAMILY=gemm16 EXTRA="--dtype bf16" bash $R/tools/profile_round.sh r04b_bf16 > $R/gpurun_out/prof_r04b_bf16.log 2>&1
cp $R/gpurun_out/prof_r04b_bf16/summary/* $R/gpurun_out/profiles/ 2>/dev/null
rm -rf $R/gpurun_out/prof_r04b $R/gpurun_out/prof_r04b_bf16
bash $R/tools/graph_timeline.sh > /dev/null 2>&1
cat $R/gpurun_out/timeline/summary.txt $R/gpurun_out/timeline/phases.txt > $R/gpurun_out/profiles/r04b_graph_timeline.txt
EXTRA="--dtype bf16" bash $R/tools/graph_timeline.sh > /dev/null 2>&1
cat $R/gpurun_out/timeline/summary.txt $R/gpurun_out/timeline/phases.txt > $R/gpurun_out/profiles/r04b_bf16_graph_timeline.txt
bash $R/tools/profile_configs.sh r04b > $R/gpurun_out/prof_cfg.log 2>&1
cp $R/gpurun_out/prof_cfg/r04b_* $R/gpurun_out/profiles/ 2>/dev/null
ls $R/gpurun_out/profiles
