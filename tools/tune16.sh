#!/bin/bash
# Generate mmfn_amd/tuning/gfx950_bf16.json on the GPU box: every bf16 GEMM / convolution shape of the benched configurations is
# timed over (tile, LDS stages, contraction split) the first time it is launched.  Usage (through gpurun): bash tools/tune16.sh
R=${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}
cd $R
export MMFN_AUTOTUNE16=1 MMFN_TUNING_FILE16=$R/gpurun_out/gfx950_bf16.json
# start from the committed table: only shapes it does not hold yet are timed (TUNE16_FRESH=1: time everything again)
if [ -n "$TUNE16_FRESH" ]; then rm -f $MMFN_TUNING_FILE16; else mkdir -p $R/gpurun_out; cp $R/mmfn_amd/tuning/gfx950_bf16.json $MMFN_TUNING_FILE16; fi
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from mmfn_amd import ops16
from mmfn_amd.config import GlobalConfig
from mmfn_amd.model import MMFN, MMFNImg
dev = torch.device("cuda", 0)
for cls, variant, B in ((MMFN, "vec", 32), (MMFNImg, "img", 32), (MMFN, "vec", 2), (MMFN, "vec", 1)):
    torch.manual_seed(42)
    net = cls(GlobalConfig(act_dtype="bf16"), dev)
    net.train()
    inp, gt = bench.synth_inputs(B, dev, seed=42, variant=variant)
    for _ in range(2):
        net.train_step(inp, gt)
    net.eval()
    with torch.no_grad():
        net._engine_for().forward(inp, False, None)
    torch.cuda.synchronize()
    print(variant, B, len(ops16._tuned), "shapes", flush=True)
    del net
ops16.save_tuning(os.environ["MMFN_TUNING_FILE16"])
PY
