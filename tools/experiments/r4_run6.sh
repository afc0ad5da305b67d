cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels16_gpu.py tests/test_gemm16_gpu.py -x -q 2>&1 | tail -8
 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "recomputes or layernorm or tokens" 2>&1 | tail -5
 timeout 900 python -m pytest tests/test_bf16_mode_gpu.py -x -q 2>&1 | tail -8
 timeout 900 python -m pytest tests/test_parallel_gpu.py -x -q -k "bf16_gradient_buckets or single_graph" 2>&1 | tail -8
 timeout 900 python -m pytest tests/test_trainer_gpu.py -x -q -k "single_graph_transport" 2>&1 | tail -8
 timeout 900 python -m pytest tests/test_e2e_gpu.py -x -q -k "train_step_matches or changes_no_bit" 2>&1 | tail -5) > gpurun_out/r6_tests.log 2>&1
(timeout 600 python tools/bf16_quality.py --batch 8 2>&1 | grep -v "^tap"
 timeout 600 python tools/bf16_quality.py --batch 32 2>&1 | grep -v "^tap") > gpurun_out/r6_quality.log 2>&1
timeout 600 python bench.py --config bf16 --no-cpu-baseline --no-also > gpurun_out/r6_bf16.json 2> gpurun_out/r6_bf16.err
cat gpurun_out/r6_tests.log gpurun_out/r6_quality.log
python -c "
import json
d=json.load(open('gpurun_out/r6_bf16.json'))
print('bf16', d['value'], d['ms_per_step'], d.get('loss_vs_oracle'))"
tail -3 gpurun_out/r6_bf16.err
