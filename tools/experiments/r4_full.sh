#!/bin/bash
# full GPU suite + the driver's default bench line + smoke, on the tree as committed
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/full_tests.log 2>&1; tail -5 gpurun_out/full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full_smoke.log 2>&1; tail -3 gpurun_out/full_smoke.log
python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; cat gpurun_out/full_bench.json
