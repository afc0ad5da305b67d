"""Glue shared by the oracle tests and the GPU parity tests (TEST INFRASTRUCTURE)."""
import numpy as np
import torch

from . import fixtures, preprocess
from .config import OracleConfig
from .model import OracleMMFN, l1_waypoint_loss


def forward_args(batch, variant="vec"):
    """Synthetic batch -> the reference's 8 forward() arguments (CPU fp32 tensors)."""
    fronts = torch.from_numpy(np.stack([preprocess.crop_chw(im) for im in batch["rgb_u8"].numpy()]).copy()).float()
    bev = torch.from_numpy(np.stack([preprocess.lidar_histogram(p[:, :3].numpy().astype(np.float64))
                                     for p in batch["lidar_pts"]])).contiguous()
    vm = [[batch["lane"]], [batch["lane_num"].float()], int(batch["lane_num"].max())]
    return ([fronts], [bev], [batch["map_u8"].float()], vm, [batch["radar"]], [batch["radar_adj"]],
            batch["target_point"], batch["velocity"])


def frames_args(seq_len, n_views, batch=2, variant="img", lanes=4):
    """Inputs with several frames per sample: frame j of every modality is synthetic_batch(seed=42 + j); labels, target
    point and velocity come from seed 42.  Returns (batches, forward() arguments, gt_wp)."""
    batches = [fixtures.synthetic_batch(batch, variant, seed=42 + j, lanes=lanes) for j in range(seq_len * n_views)]
    per = [forward_args(b, variant) for b in batches]
    first = per[0]
    args = ([a[0][0] for a in per], [a[1][0] for a in per[:seq_len]], [a[2][0] for a in per[:seq_len]],
            first[3], first[4], first[5], first[6], first[7])
    return batches, args, batches[0]["gt_wp"]


def build_oracle(variant="vec", dropout=0.0, **cfg_kw):
    cfg = OracleConfig(embd_pdrop=dropout, attn_pdrop=dropout, resid_pdrop=dropout, **cfg_kw)
    model = OracleMMFN(cfg, "cpu", variant)
    fixtures.fill_module(model)
    return model


def calibrate_bn(model, args):
    """One train-mode forward with BN momentum 1.0: running stats := batch stats."""
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(*args)
    for m in bns:
        m.momentum = 0.1
    model.eval()


def train_step(model, args, gt_wp, lr=1e-4):
    """zero-grad + forward + L1 + backward + AdamW (phase2_train_net.py:60-110)."""
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    for p in model.parameters():
        p.grad = None
    pred = model(*args)
    loss = l1_waypoint_loss(pred, gt_wp)
    loss.backward()
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}
    opt.step()
    return pred.detach(), loss.detach(), grads
