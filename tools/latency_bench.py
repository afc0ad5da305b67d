"""Batch-1 closed-loop latency of mmfn_amd.inference.DrivingSession (hipGraph vs eager), MI355X."""
import json
import sys
import time
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmfn_amd.config import GlobalConfig  # noqa: E402
from mmfn_amd.inference import DrivingSession  # noqa: E402
from mmfn_amd.model import MMFN  # noqa: E402


def main():
    net = MMFN(GlobalConfig(), "cuda:0")
    rng = np.random.RandomState(0)
    rgb = rng.randint(0, 256, (300, 400, 3)).astype(np.uint8)
    pts = np.stack([rng.uniform(-20, 20, 16384), rng.uniform(-12, 28, 16384), rng.uniform(-3, 1, 16384), rng.uniform(0, 1, 16384)], 1).astype(np.float32)
    lanes = rng.randn(40, 10, 5).astype(np.float32)
    out = {}
    eng = net._engine_for()
    modes = (("hipgraph", True, True, True), ("hipgraph_batchnorm_not_folded", True, True, False), ("hipgraph_single_stream", True, False, True),
             ("eager", False, True, True))
    for name, use_graph, multi, fold in modes:
        eng.multi_stream = multi
        sess = DrivingSession(net, use_graph=use_graph, fold_batchnorm=fold)
        for _ in range(5):
            sess.predict(rgb, pts, lanes, (3.0, 20.0), 4.0)
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            sess.predict(rgb, pts, lanes, (3.0, 20.0), 4.0)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        t0 = time.perf_counter()
        for _ in range(20):
            sess._load(rgb, pts, lanes, (3.0, 20.0), 4.0)
        torch.cuda.synchronize()
        load_ms = (time.perf_counter() - t0) / 20 * 1e3
        out[name] = {"min_ms": round(ts[0], 3), "median_ms": round(ts[len(ts) // 2], 3), "p90_ms": round(ts[int(len(ts) * 0.9)], 3),
                     "max_ms": round(ts[-1], 3), "stage_inputs_ms": round(load_ms, 3)}
    eng.multi_stream = True
    print(json.dumps({"workload": "batch-1 tick: 300x400 u8 frame + 32768-pt sweep + 40 lanes -> waypoints (host to host)", **out}))


if __name__ == "__main__":
    main()
