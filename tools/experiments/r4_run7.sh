cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
(timeout 1500 python -m pytest tests/test_bf16_mode_gpu.py -x -q -s 2>&1 | grep -v "^$" | grep "bf16 mode\|autocast\|loss:\|passed\|failed\|Error\|assert" | cut -c1-400) > gpurun_out/r7_tests.log 2>&1
timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "accumulates or fused_train_step" 2>&1 | tail -4 >> gpurun_out/r7_tests.log
cat gpurun_out/r7_tests.log
# fp32 profile of the round
timeout 1500 bash tools/profile_round.sh r04a > gpurun_out/r7_prof.log 2>&1
cp gpurun_out/prof_r04a/summary/* gpurun_out/profiles/ 2>/dev/null
rm -rf gpurun_out/prof_r04a
timeout 600 bash tools/graph_timeline.sh > /dev/null 2>&1
cp gpurun_out/timeline/summary.txt gpurun_out/profiles/r04a_graph_timeline.txt; rm -rf gpurun_out/timeline
# bf16 mode
EXTRA="--dtype bf16" timeout 1500 bash tools/profile_round.sh r04a_bf16 > gpurun_out/r7_prof16.log 2>&1
cp gpurun_out/prof_r04a_bf16/summary/* gpurun_out/profiles/ 2>/dev/null
rm -rf gpurun_out/prof_r04a_bf16
EXTRA="--dtype bf16" timeout 600 bash tools/graph_timeline.sh > /dev/null 2>&1
cp gpurun_out/timeline/summary.txt gpurun_out/profiles/r04a_bf16_graph_timeline.txt; rm -rf gpurun_out/timeline
ls -la gpurun_out/profiles
head -12 gpurun_out/profiles/r04a_graph_timeline.txt
head -12 gpurun_out/profiles/r04a_bf16_graph_timeline.txt
cat gpurun_out/profiles/r04a_traffic.json 2>/dev/null | head -30
