"""End-to-end rate of the real training loop (phase-1 frames -> loader -> pinned staging -> H2D -> fused step): pickled PRE_Data
frames through DataLoader workers with eager launches / static-input hipGraph replay, and the packed store (data.pack_frames +
PackedLoader) with graph replay.  Complements bench.py, whose inputs are resident in HBM."""
import json
import os
import pickle
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmfn_amd import data as D  # noqa: E402
from mmfn_amd.config import GlobalConfig  # noqa: E402
from mmfn_amd.model import MMFN  # noqa: E402
from mmfn_amd.optim import FusedAdamW  # noqa: E402
from mmfn_amd.trainer import Trainer  # noqa: E402


def write_frames(root, n, seed=0):
    rng = np.random.RandomState(seed)
    for i in range(n):
        radar = rng.randn(81, 5)
        s = {"fronts": [torch.from_numpy(rng.randint(0, 256, (3, 256, 256)).astype(np.uint8))],
             "lidars": [(rng.randint(0, 6, (2, 256, 256)) / 5.0).astype(np.float32)],
             "vectormaps": [torch.from_numpy(rng.randn(int(rng.randint(49, 65)), 10, 5))],
             "radar": [radar], "maps": [torch.from_numpy(rng.randint(0, 256, (3, 256, 256)).astype(np.uint8))],
             "waypoints": [tuple(rng.randn(2)) for _ in range(5)], "target_point": tuple(rng.randn(2) * 10.0),
             "steer": 0.0, "throttle": 0.5, "brake": False, "command": 1, "velocity": float(rng.uniform(0, 8))}
        with open(os.path.join(root, "%d.pkl" % i), "wb") as fd:
            pickle.dump(s, fd)


def run(mode, B=32, n=768, epochs=2):
    cfg = GlobalConfig()
    with tempfile.TemporaryDirectory() as tmp:
        write_frames(tmp, n)
        store = D.PRE_Data(tmp, cfg, "train")
        torch.manual_seed(0)
        net = MMFN(cfg, "cuda:0")
        opt = FusedAdamW(net, lr=1e-4)
        tr = Trainer("cuda:0", None)
        if mode == "packed":   # the flat memory-mapped store (data.pack_frames, one-time conversion) + its threaded loader
            t_pack = time.time()
            packed = D.PackedFrames(D.pack_frames(store, os.path.join(tmp, "packed")))
            t_pack = time.time() - t_pack
            loader = D.PackedLoader(packed, batch_size=B)
        else:
            loader = torch.utils.data.DataLoader(store, batch_size=B, shuffle=False, num_workers=8, collate_fn=D.collate,
                                                 persistent_workers=True, prefetch_factor=4)
        graph = mode != "eager"
        tr.train(net, loader, cfg, opt, graph=graph)  # warm epoch (buffers, capture, worker start-up)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(epochs):
            tr.train(net, loader, cfg, opt, graph=graph)
        torch.cuda.synchronize()
        dt = time.time() - t0
        res = {"samples_per_s": round(epochs * n / dt, 1), "ms_per_step": round(dt / (epochs * n / B) * 1e3, 2),
               "host_threads": torch.get_num_threads()}
        # what the input side alone sustains (loader + pinned staging + H2D, no training step)
        t1 = time.time()
        for _ in D.DevicePrefetcher(loader, "cuda:0", cfg):
            pass
        torch.cuda.synchronize()
        res["input_side_only_samples_per_s"] = round(n / (time.time() - t1), 1)
        if mode == "packed":
            res["pack_seconds_per_1000_frames"] = round(t_pack / n * 1000, 2)
        return res


def main():
    if len(sys.argv) > 1:  # child: one mode per process (loader workers and 20 GB of engine buffers do not pile up)
        print(json.dumps(run(sys.argv[1])))
        return
    import subprocess
    out = {}
    for mode in ("eager", "graph", "packed"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True)
        out[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps({"workload": "Trainer.train, batch 32, 768 phase-1 frames/epoch, one MI355X; eager / graph: pickles through 8 persistent "
                                  "DataLoader workers; packed: data.PackedLoader over the memory-mapped conversion of the same frames, "
                                  "graph replay", **out}))


if __name__ == "__main__":
    main()
