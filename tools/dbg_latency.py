import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmfn_amd.config import GlobalConfig
from mmfn_amd.inference import DrivingSession
from mmfn_amd.model import MMFN
net = MMFN(GlobalConfig(), "cuda:0")
rng = np.random.RandomState(0)
rgb = rng.randint(0, 256, (300, 400, 3)).astype(np.uint8)
pts = rng.uniform(-20, 20, (16384, 4)).astype(np.float32)
lanes = rng.randn(40, 10, 5).astype(np.float32)
sess = DrivingSession(net)
for _ in range(5):
    sess.predict(rgb, pts, lanes, (3.0, 20.0), 4.0)
rows = []
for _ in range(40):
    t0 = time.perf_counter()
    lidar = np.asarray(pts)
    sweep = np.append(lidar, sess.prev_sweep, axis=0)
    sess.prev_sweep = lidar
    t1 = time.perf_counter()
    sess._load(rgb, sweep, lanes, (3.0, 20.0), 4.0)
    t2 = time.perf_counter()
    sess.graph.replay()
    t3 = time.perf_counter()
    sess.out_host.copy_(sess.pred, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    t4 = time.perf_counter()
    rows.append([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4))])
for r in rows:
    print(" ".join("%7.3f" % x for x in r))
