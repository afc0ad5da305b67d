mkdir -p gpurun_out/r06
(time timeout 1200 python -m pytest tests -m gpu -x -q) > gpurun_out/r06/gputest.log 2>&1
timeout 400 python bench.py > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err
bash tools/profile_round.sh r06 > gpurun_out/r06/prof.log 2>&1
EXTRA="--dtype bf16" bash tools/profile_round.sh r06_bf16 > gpurun_out/r06/prof_bf16.log 2>&1
cp gpurun_out/prof_r06/summary/* gpurun_out/prof_r06_bf16/summary/* gpurun_out/r06/
rm -rf gpurun_out/prof_r06 gpurun_out/prof_r06_bf16
bash tools/graph_timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline/summary.txt gpurun_out/r06/r06_graph_timeline.txt
EXTRA="--dtype bf16" bash tools/graph_timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline/summary.txt gpurun_out/r06/r06_bf16_graph_timeline.txt
rm -rf gpurun_out/timeline
timeout 300 python tools/phase_clock.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/r06_phase_clock_f32.txt
timeout 300 python tools/phase_clock.py --dtype bf16 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/r06_phase_clock_bf16.txt
(timeout 300 python tools/phase_clock.py --lanes; timeout 300 python tools/phase_clock.py --lanes --dtype bf16) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/r06_lane_clock.txt
bash tools/attn_profile.sh r06 > /dev/null 2>&1; cp gpurun_out/attn_r06/summary.txt gpurun_out/r06/r06_attn_pmc.txt; rm -rf gpurun_out/attn_r06
bash tools/profile_configs.sh r06 > gpurun_out/r06/prof_cfg.log 2>&1; cp gpurun_out/prof_cfg/r06_* gpurun_out/r06/; rm -rf gpurun_out/prof_cfg
tail -3 gpurun_out/r06/gputest.log; cut -c1-600 gpurun_out/r06/bench.json; ls gpurun_out/r06
