"""MMFN model class with the reference's constructor / forward / control_pid / checkpoint contract
(mmfn_utils/models/model_vec.py:626-726, model_img.py:451-550, model_rad.py:656-745), executing on
hand-written gfx950 kernels through mmfn_amd.engine.

Drop-in surface:
  MMFN(config, device)                                   same ctor
  forward(image_list, lidar_list, maps_list, vectormaps_list, radar_list, radar_adj,
          target_point, velocity) -> pred_wp [B, pred_len, 2]
  control_pid(waypoints, velocity)                       CPU numpy PID, as the agents call it
  state_dict()/load_state_dict()                         identical keys / shapes / order
  parameters()                                           reference order (optimizer state files map 1:1)
Training through autograd works (`loss.backward()` fills p.grad from the flat gradient buffer);
`train_step()` is the fused fast path (forward + L1 + backward + AdamW, hipGraph-capturable).
"""
from collections import deque

import numpy as np
import torch
import torch.nn as nn

from . import params as P
from ._lib import MMFNLibraryError


class PIDController(object):
    """model_vec.py:601-623."""

    def __init__(self, K_P=1.0, K_I=0.0, K_D=0.0, n=20):
        self._K_P, self._K_I, self._K_D = K_P, K_I, K_D
        self._window = deque([0 for _ in range(n)], maxlen=n)
        self._max = 0.0
        self._min = 0.0

    def step(self, error):
        self._window.append(error)
        self._max = max(self._max, abs(error))
        self._min = -abs(self._max)
        if len(self._window) >= 2:
            integral = np.mean(self._window)
            derivative = self._window[-1] - self._window[-2]
        else:
            integral, derivative = 0.0, 0.0
        return self._K_P * error + self._K_I * integral + self._K_D * derivative


class _AutogradBridge(torch.autograd.Function):
    """Lets a reference-style loop (`loss = f(model(...)); loss.backward()`) drive the explicit
    HIP backward: the whole network is one autograd node."""

    @staticmethod
    def forward(ctx, anchor, module, inp):
        pred, _ = module._engine_for().forward(inp, True, None)
        ctx.module = module
        return pred.clone()

    @staticmethod
    def backward(ctx, dpred):
        m = ctx.module
        L = m._layout
        # torch semantics: backward ACCUMULATES into a .grad that exists and writes one that does not (after
        # optimizer.zero_grad() / p.grad = None, phase2_train_net.py:60).  The explicit backward overwrites the flat gradient
        # buffer, so when the previous gradients are still attached they are parked and added back (two extra passes over the
        # buffer, on this path only; the fused Engine.train_step never accumulates)
        held = next((p.grad for n, p in m.named_parameters() if n not in L.unused), None)
        accumulate = held is not None and held.data_ptr() >= L.grads.data_ptr() and \
            held.data_ptr() < L.grads.data_ptr() + L.grads.numel() * 4
        if accumulate:
            if getattr(L, "grads_parked", None) is None:
                L.grads_parked = torch.empty_like(L.grads)
            L.grads_parked.copy_(L.grads)
        m._engine_for().backward(dpred.contiguous(), 1.0)
        if accumulate:
            from . import ops
            ops.axpby(L.grads[:L.tail], L.grads_parked[:L.tail], 1.0, 1.0)
        L.attach_grads()
        m.weights_changed()   # an optimizer step on p.grad follows
        return None, None, None


class MMFN(nn.Module):
    """Transformer-based multi-modal fusion + GRU waypoint head (vec variant by default)."""

    variant = "vec"

    def __init__(self, config, device, variant=None):
        super().__init__()
        if variant is not None:
            self.variant = variant
        self.device = device
        self.config = config
        self.pred_len = config.pred_len
        self.turn_controller = PIDController(config.turn_KP, config.turn_KI, config.turn_KD, config.turn_n)
        self.speed_controller = PIDController(config.speed_KP, config.speed_KI, config.speed_KD, config.speed_n)
        self.encoder = P.encoder_params(config, self.variant)
        self.join = nn.Sequential(nn.Linear(512, 256), nn.ReLU(inplace=True), nn.Linear(256, 128), nn.ReLU(inplace=True),
                                  nn.Linear(128, 64), nn.ReLU(inplace=True))
        self.decoder = nn.GRUCell(input_size=2, hidden_size=64)
        self.output = nn.Linear(64, 2)
        object.__setattr__(self, "_layout", P.FlatLayout(self, P.default_unused(self.variant)))
        object.__setattr__(self, "_engine", None)
        object.__setattr__(self, "_anchor", None)
        self._layout.materialize(device)

    # ------------------------------------------------------------------ device moves keep the flat layout
    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        dev = next(self.parameters()).device
        self._layout.materialize(dev)
        object.__setattr__(self, "_engine", None)
        self.device = dev
        return self

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.weights_changed()
        return out

    def weights_changed(self):
        """Note that parameters / BatchNorm statistics changed (the engine's optimizer step, a graph replay of it, the autograd
        bridge's backward and load_state_dict call this): consumers of tensors derived from the weights - the BatchNorm-folded
        filters of inference.DrivingSession - compare the counter and re-derive."""
        object.__setattr__(self, "_weights_version", getattr(self, "_weights_version", 0) + 1)

    def _engine_for(self):
        if self._engine is None:
            from .engine import Engine
            if self._layout.device.type != "cuda":
                raise MMFNLibraryError("MMFN runs on hand-written HIP kernels only: construct it on a GPU device "
                                       "(got %s); there is no CPU fallback" % self._layout.device)
            object.__setattr__(self, "_engine", Engine(self, self._layout, self.variant))
            object.__setattr__(self, "_anchor", torch.zeros(1, device=self._layout.device, requires_grad=True))
        return self._engine

    # ------------------------------------------------------------------ reference forward signature
    def _pack(self, image_list, lidar_list, maps_list, vectormaps_list, radar_list, radar_adj, target_point, velocity):
        dev = self._layout.device
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        cfg = self.config
        S = cfg.seq_len
        if len(lidar_list) != S or not image_list or len(image_list) % S:
            raise ValueError("expected seq_len = %d LiDAR frames and a multiple of that many camera frames, got %d and %d"
                             % (S, len(lidar_list), len(image_list)))
        cfg.n_views = len(image_list) // S  # the reference mutates config here (model_vec.py:504)
        tokens = self.encoder.transformer1.pos_emb.shape[1]
        if (cfg.n_views + 2) * S * 64 != tokens:
            # the reference fails on the same input, later: pos_emb + token_embeddings do not broadcast (model_vec.py:235)
            raise ValueError("%d camera frames per sample, but the position embeddings were built for n_views = %d"
                             % (len(image_list), tokens // (64 * S) - 2))
        # frames of one sample are consecutive batch entries: torch.stack(list, dim=1).view(bz * n, ...) (model_vec.py:506-508)
        frames = lambda lst: f32(lst[0]) if len(lst) == 1 else torch.stack([f32(t) for t in lst], dim=1).flatten(0, 1)
        inp = {"image": frames(image_list), "lidar": frames(lidar_list), "target_point": f32(target_point),
               "velocity": f32(velocity).view(-1)}
        assert inp["image"].shape[2:] == inp["lidar"].shape[2:], "image and LiDAR BEV must share H x W (model_vec.py:500,506)"
        if self.variant == "img":
            if len(maps_list) != S:
                raise ValueError("expected seq_len = %d map frames, got %d" % (S, len(maps_list)))
            inp["map"] = frames(maps_list)
        else:
            lane = vectormaps_list[0][0]
            lane_num = vectormaps_list[1][0]
            if lane.dim() == 5:  # agent packing [1,1,L,10,5] (e2e_agent/mmfn_vectornet.py:287-293)
                lane = lane[0]
            B = lane.shape[0]
            inp["lane"] = f32(lane)
            inp["lane_num"] = lane_num.reshape(B).to(device=dev, dtype=torch.int32).contiguous()
        if self.variant == "rad":
            inp["radar"] = f32(radar_list[0]).view(-1, 81, 5)
            inp["radar_adj"] = f32(radar_adj[0]).view(-1, 81, 81)
        return inp

    def forward(self, image_list, lidar_list, maps_list, vectormaps_list, radar_list, radar_adj, target_point, velocity):
        inp = self._pack(image_list, lidar_list, maps_list, vectormaps_list, radar_list, radar_adj, target_point, velocity)
        eng = self._engine_for()
        if self.training and torch.is_grad_enabled():
            return _AutogradBridge.apply(self._anchor, self, inp)
        pred, _ = eng.forward(inp, self.training, None)
        return pred.clone()

    # ------------------------------------------------------------------ fused fast path
    def train_step(self, inp, gt_wp, lr=1e-4, dp=None):
        """One full training step on device-resident inputs (see engine.Engine.train_step)."""
        return self._engine_for().train_step(inp, gt_wp, lr=lr, dp=dp)

    # ------------------------------------------------------------------ PID (model_vec.py:684-726)
    def control_pid(self, waypoints, velocity):
        assert waypoints.size(0) == 1
        wp = waypoints[0].data.cpu().numpy()
        wp[:, 1] *= -1
        speed = velocity[0].data.cpu().numpy()
        desired_speed = np.linalg.norm(wp[0] - wp[1]) * 2.0
        brake = desired_speed < self.config.brake_speed or (speed / desired_speed) > self.config.brake_ratio
        aim = (wp[1] + wp[0]) / 2.0
        angle = np.degrees(np.pi / 2 - np.arctan2(aim[1], aim[0])) / 90
        if speed < 0.01:
            angle = np.array(0.0)
        steer = np.clip(self.turn_controller.step(angle), -1.0, 1.0)
        delta = np.clip(desired_speed - speed, 0.0, self.config.clip_delta)
        throttle = np.clip(self.speed_controller.step(delta), 0.0, self.config.max_throttle)
        throttle = throttle if not brake else 0.0
        metadata = {
            "speed": float(speed.astype(np.float64)), "steer": float(steer), "throttle": float(throttle),
            "brake": float(brake), "wp_2": tuple(wp[1].astype(np.float64)), "wp_1": tuple(wp[0].astype(np.float64)),
            "desired_speed": float(desired_speed.astype(np.float64)), "angle": float(angle.astype(np.float64)),
            "aim": tuple(aim.astype(np.float64)), "delta": float(delta.astype(np.float64)),
        }
        return steer, throttle, brake, metadata


class MMFNImg(MMFN):
    variant = "img"


class MMFNRad(MMFN):
    variant = "rad"
