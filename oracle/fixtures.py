"""Deterministic parameter fill + seeded synthetic batches (TEST INFRASTRUCTURE).

Shared by the reference-import script (oracle/make_golden.py), the oracle tests
and the GPU parity tests so that reference, oracle and HIP product all see the
same weights and inputs without shipping weight files.  Closed-form recipe per
SURVEY.md section 8c / section 9 "Fixture-strategy sanity".
"""
import zlib

import numpy as np
import torch

PHI = 0.6180339887498949


def _wave(key, n):
    i = np.arange(n, dtype=np.float64)
    phase = float(zlib.crc32(key.encode()) % 100003) * 1e-3
    return np.sin(i * (PHI * 7.0) + phase)


def fill_value(key, shape, dtype=torch.float32):
    """Closed-form value for state_dict entry ``key`` of ``shape``."""
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    u = _wave(key, n)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "running_var":
        v = 1.0 + 0.25 * np.abs(u)
    elif leaf == "running_mean":
        v = 0.1 * u
    elif leaf == "pos_emb":
        v = 0.1 * u
    elif len(shape) >= 2:
        # GAT params are (in, out); everything else is (out, in, ...)
        fan_in = shape[0] if leaf in ("W", "a") else int(np.prod(shape[1:]))
        v = u * np.sqrt(3.0 / fan_in)
    elif leaf == "weight":  # 1-D weight => a norm layer's gamma
        v = 1.0 + 0.1 * u
    else:  # biases
        v = 0.1 * u
    return torch.from_numpy(v.reshape(shape)).to(dtype)


@torch.no_grad()
def fill_module(module):
    """Overwrite every parameter/buffer of ``module`` with the closed-form fill."""
    sd = module.state_dict()
    for k, t in sd.items():
        t.copy_(fill_value(k, t.shape).to(t.dtype))
    return module


def synthetic_batch(batch, variant="vec", seed=42, n_lidar=16384, lanes=64, ragged=True):
    """Seeded synthetic inputs in the collated layout.

    Returns a dict of CPU tensors:
      rgb_u8 [B,300,400,3] u8 HWC camera frames, lidar_pts [B,N,4] f32 XYZI,
      lane [B,L,10,5] f32 (rows >= lane_num zero, as pad_sequence leaves them),
      lane_num [B] i64, map_u8 [B,3,256,256] u8 (img), radar [B,81,5], radar_adj [B,81,81],
      target_point [B,2], velocity [B], gt_wp [B,4,2].
    """
    g = torch.Generator().manual_seed(seed)
    b = batch
    out = {}
    out["rgb_u8"] = torch.randint(0, 256, (b, 300, 400, 3), generator=g, dtype=torch.uint8)
    pts = torch.empty(b, n_lidar, 4)
    pts[..., 0:2] = torch.rand(b, n_lidar, 2, generator=g) * 40.0 - 20.0
    pts[..., 2] = torch.rand(b, n_lidar, generator=g) * 4.0 - 3.0
    pts[..., 3] = torch.rand(b, n_lidar, generator=g)
    n_pad = n_lidar // 16  # tail padding placed far out of range -> dropped by the histogram
    pts[:, n_lidar - n_pad:, 0] = 1e6
    out["lidar_pts"] = pts
    lane = torch.zeros(b, lanes, 10, 5)
    lane[..., 0:2] = torch.randn(b, lanes, 10, 2, generator=g) * 8.0
    lane[..., 2:5] = torch.randint(0, 2, (b, lanes, 10, 3), generator=g).float()
    if ragged:
        lane_num = torch.randint(1, lanes + 1, (b,), generator=g)
        lane_num[0] = lanes  # pad_sequence pads to the longest sample
    else:
        lane_num = torch.full((b,), lanes, dtype=torch.int64)
    for i in range(b):
        lane[i, int(lane_num[i]):] = 0.0
    out["lane"] = lane
    out["lane_num"] = lane_num
    out["map_u8"] = torch.randint(0, 256, (b, 3, 256, 256), generator=g, dtype=torch.uint8)
    radar = torch.randn(b, 81, 5, generator=g)
    radar[..., 3] = radar[..., 3].abs() + 0.5
    out["radar"] = radar
    out["radar_adj"] = radar[:, None, :, 1] - radar[:, :, None, 1]  # adj[i,j] = r[j,1]-r[i,1]
    out["target_point"] = torch.randn(b, 2, generator=g) * 10.0
    out["velocity"] = torch.rand(b, generator=g) * 8.0
    out["gt_wp"] = torch.randn(b, 4, 2, generator=g) * 5.0
    return out


def synthetic_samples(lane_counts=(5, 9, 3), seed=11, radar_counts=(50, 100, 81)):
    """Per-frame dicts in the schema phase 1 pickles (CARLA_Data.__getitem__, dataloader.py:183-268):
    u8 camera crop, f32 BEV histogram, f64 lanes [L,10,5], radar forced to [81,5], u8 raster map,
    seq_len+pred_len local waypoints, target point and the scalar labels."""
    rng = np.random.RandomState(seed)
    out = []
    for n_lane, n_rad in zip(lane_counts, radar_counts):
        radar = np.zeros((81, 5))
        m = min(n_rad, 81)
        radar[:m] = rng.randn(m, 5)
        out.append({
            "fronts": [torch.from_numpy(rng.randint(0, 256, (3, 256, 256)).astype(np.uint8))],
            "lidars": [(rng.randint(0, 6, (2, 256, 256)) / 5.0).astype(np.float32)],
            "vectormaps": [torch.from_numpy(rng.randn(n_lane, 10, 5))],
            "radar": [radar],
            "maps": [torch.from_numpy(rng.randint(0, 256, (3, 256, 256)).astype(np.uint8))],
            "waypoints": [tuple(rng.randn(2)) for _ in range(5)],
            "target_point": tuple(rng.randn(2) * 10.0),
            "steer": float(rng.uniform(-1, 1)), "throttle": float(rng.uniform(0, 0.75)), "brake": bool(n_lane % 2),
            "command": int(rng.randint(1, 5)), "velocity": float(rng.uniform(0, 8)),
        })
    return out


def write_synthetic_route(root, seed=21, n_frames=9, route="route00"):
    """A recorded-route folder in the layout CARLA_Data scans (dataloader.py:64-139): rgb_front/ maps/ lidar/ radar/
    vectormap/ measurements/ with NNNN.{png,npy,json}, frames 0000..n_frames-1.  Deterministic in `seed`."""
    import json
    import os
    from PIL import Image
    rng = np.random.RandomState(seed)
    d = os.path.join(root, route)
    for sub in ("rgb_front", "maps", "lidar", "radar", "vectormap", "measurements"):
        os.makedirs(os.path.join(d, sub), exist_ok=True)
    x = y = 0.0
    theta = 0.3
    for i in range(n_frames):
        k = "%04d" % i
        Image.fromarray(rng.randint(0, 256, (300, 400, 3)).astype(np.uint8)).save(os.path.join(d, "rgb_front", k + ".png"))
        Image.fromarray(rng.randint(0, 256, (256, 256, 3)).astype(np.uint8)).save(os.path.join(d, "maps", k + ".png"))
        n = 3000 + 100 * i
        pts = np.stack([rng.uniform(-20, 20, n), rng.uniform(-12, 28, n), rng.uniform(-3, 1, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
        np.save(os.path.join(d, "lidar", k + ".npy"), pts)
        rad = rng.randn(40 + 15 * i, 5)
        rad[:, 3] = np.abs(rad[:, 3]) + 0.5
        np.save(os.path.join(d, "radar", k + ".npy"), rad)
        np.save(os.path.join(d, "vectormap", k + ".npy"), rng.randn(3 + i, 10, 5))
        x, y, theta = x + rng.uniform(0.5, 1.5), y + rng.uniform(-0.3, 0.3), theta + rng.uniform(-0.05, 0.05)
        meas = {"x": x, "y": y, "theta": theta, "x_command": x + 20.0, "y_command": y - 5.0, "steer": float(rng.uniform(-1, 1)),
                "throttle": float(rng.uniform(0, 0.75)), "brake": bool(i % 3 == 0), "command": int(rng.randint(1, 5)),
                "speed": float(rng.uniform(0, 8))}
        with open(os.path.join(d, "measurements", k + ".json"), "w") as f:
            json.dump(meas, f)
    return d
