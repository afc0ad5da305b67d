"""Generate mmfn_amd/tuning/gfx950.json: run eager training steps of the bench workloads with MMFN_AUTOTUNE=1 so every
GEMM / conv shape of the step is timed once over the (tile, split-K) grid.  Run on the GPU box:
    MMFN_AUTOTUNE=1 python tools/tune.py gpurun_out/gfx950.json      then copy the file into mmfn_amd/tuning/."""
import os
import sys
import time

os.environ["MMFN_AUTOTUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mmfn_amd import ops  # noqa: E402
from mmfn_amd.config import GlobalConfig  # noqa: E402
from mmfn_amd.model import MMFN, MMFNImg, MMFNRad  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gfx950.json"
    dev = torch.device("cuda:0")
    jobs = [("vec", 32, 16384, False), ("img", 32, 16384, False), ("rad", 16, 65536, False), ("vec", 128, 16384, True)]
    dtype = os.environ.get("TUNE_DTYPE", "f32")  # "bf16": tune the bf16-operand kernels' (tile, split-K) as well
    if os.environ.get("TUNE_JOBS"):
        jobs = jobs[:int(os.environ["TUNE_JOBS"])]
    for variant, B, n_lidar, image_only in jobs:
        t0 = time.time()
        torch.manual_seed(42)
        net = {"vec": MMFN, "img": MMFNImg, "rad": MMFNRad}[variant](GlobalConfig(gemm_dtype=dtype), dev)
        net.train()
        inp, gt = bench.synth_inputs(B, dev, seed=42, n_lidar=n_lidar, variant=variant)
        eng = net._engine_for()
        eng.multi_stream = False
        if image_only:
            step = bench.ImageBranchOnly(eng, inp["rgb_u8"])
            step()
        else:
            eng.train_step(inp, gt)
        torch.cuda.synchronize()
        print("%s B=%d%s: %d shapes tuned so far (%.0f s)" % (variant, B, " image-only" if image_only else "", len(ops._tuned), time.time() - t0), flush=True)
        del net, eng, inp, gt
        torch.cuda.empty_cache()
        ops.save_tuning(out)
    changed = sum(1 for v in ops._tuned.values() if v != (0, 0))
    print("wrote %s: %d shapes, %d with a non-default choice" % (out, len(ops._tuned), changed))


if __name__ == "__main__":
    main()
