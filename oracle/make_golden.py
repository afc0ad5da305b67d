"""Generate tests/golden/*.npz by running the REFERENCE itself (authoring container only).

TEST INFRASTRUCTURE.  Imports /root/reference/team_code/mmfn_utils read-only at run time
(nothing is copied) and records inputs-recipe -> outputs vectors that pin oracle/model.py
and, through it, the HIP product.  Skips with a message when /root/reference is absent
(e.g. on the GPU box).

Two import shims are needed because the container lacks the reference's pinned third-party
stack (Dockerfile:41: torch 1.10.2 / torchvision 0.11.3):
  * ``torchvision.models`` -> the BasicBlock ResNet-18/34 from oracle/model.py (torchvision is a
    third-party dependency, not reference code; its published architecture is restated there),
  * ``torch._six.string_classes`` -> (str, bytes) for data_utils.py:5.
Everything MMFN-specific (GPT, VectorNet, GAT, Encoder orchestration, GRU head, collate,
histogram, crop, radar_to_size, PID) executes the reference's own source.

Usage:  python oracle/make_golden.py
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import fixtures  # noqa: E402
from oracle.model import _ResNetTrunk  # noqa: E402


def install_shims():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.resnet34 = lambda pretrained=False, **kw: _ResNetTrunk((3, 4, 6, 3), 3)
    tvm.resnet18 = lambda pretrained=False, **kw: _ResNetTrunk((2, 2, 2, 2), 3)
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    six = types.ModuleType("torch._six")
    six.string_classes = (str, bytes)
    sys.modules["torch._six"] = six
    sys.path.insert(0, os.path.join(REF, "team_code"))


def reference_inputs(batch, ref_dl):
    """Turn a synthetic batch into the reference's forward() arguments using ITS preprocessing."""
    rgb = batch["rgb_u8"].numpy()

    class _Img:  # duck-typed PIL image for scale_and_crop_image (dataloader.py:296-308)
        def __init__(self, arr):
            self.arr, self.height, self.width = arr, arr.shape[0], arr.shape[1]

        def resize(self, size):
            return self.arr

    fronts = np.stack([ref_dl.scale_and_crop_image(_Img(im), scale=1, crop=256) for im in rgb])
    bev = np.stack([ref_dl.lidar_to_histogram_features(p[:, :3].numpy().astype(np.float64), crop=256)
                    for p in batch["lidar_pts"]])
    # contiguous(): the reference's histogram comes back channels-last strided (np.transpose +
    # astype keeps 'K' order); oneDNN then takes an NHWC conv path whose rounding differs by ~2e-6.
    # The collated training batch is plain NCHW, so pin the vectors on that layout.
    return torch.from_numpy(fronts.copy()).float(), torch.from_numpy(bev).float().contiguous()


def run_variant(variant, ref_models, ref_dl, ref_cfg, out_dir):
    torch.manual_seed(0)
    cfg = ref_cfg.GlobalConfig()
    model = getattr(ref_models, "model_" + variant).MMFN(cfg, "cpu")
    fixtures.fill_module(model)
    keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    pnames = [k for k, _ in model.named_parameters()]

    b = 2
    batch = fixtures.synthetic_batch(b, variant, seed=42, n_lidar=16384, lanes=9 if variant != "img" else 4)
    fronts, bev = reference_inputs(batch, ref_dl)
    maps = batch["map_u8"].float()
    vm = [[batch["lane"]], [batch["lane_num"].float()], int(batch["lane_num"].max())]
    args = ([fronts], [bev], [maps], vm, [batch["radar"]], [batch["radar_adj"]],
            batch["target_point"], batch["velocity"])

    res = {"bev": bev.numpy(), "fronts_crop_sum": np.float64(fronts.double().sum().item())}
    # ---- disable dropout everywhere (the reference's RNG stream cannot be matched) ----
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0
    # ---- eval forward.  The closed-form running stats are far from the statistics of raw 0..255
    # inputs, so eval mode would be ill-conditioned (|wp| ~ 1e13).  Calibrate first: one train-mode
    # forward with BatchNorm momentum 1.0 makes running stats == batch stats; then eval. ----
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(*args)
    model.eval()
    with torch.no_grad():
        res["eval_pred_wp"] = model(*args).numpy()
        res["eval_loss"] = np.float32(torch.nn.functional.l1_loss(
            torch.from_numpy(res["eval_pred_wp"]), batch["gt_wp"], reduction="none").mean().item())
    # agent-style vectormap packing (e2e_agent/mmfn_vectornet.py:287-293): data[2] is a tensor
    if variant != "img":
        with torch.no_grad():
            one = [[batch["lane"][:1]], [batch["lane_num"][:1].int()], batch["lane_num"][:1].int().view(1, 1)]
            a1 = ([fronts[:1]], [bev[:1]], None, one, [batch["radar"][:1]], [batch["radar_adj"][:1]],
                  batch["target_point"][:1], batch["velocity"][:1])
            res["eval_pred_wp_b1_agent"] = model(*a1).numpy()
    else:   # the image-map agent's call (e2e_agent/mmfn_imgnet.py:273-276): batch 1, the raster in the maps list, no vector map / radar
        with torch.no_grad():
            res["eval_pred_wp_b1_agent"] = model([fronts[:1]], [bev[:1]], [maps[:1]], None, None, None,
                                                 batch["target_point"][:1], batch["velocity"][:1]).numpy()
    for m in bns:
        m.momentum = 0.1
    fixtures.fill_module(model)  # restore closed-form running stats for the train-step vectors

    # ---- one full train step (dropout disabled above) ----
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)  # phase2_train_net.py:256
    taps = {}
    hooks = []
    enc = model.encoder
    hooks.append(enc.register_forward_hook(lambda m, i, o: taps.__setitem__("fused", o.detach().clone())))
    if variant != "img":
        hooks.append(enc.vectornet_encoder.register_forward_hook(
            lambda m, i, o: taps.__setitem__("vectornet", o.detach().clone())))
    for i in range(1, 5):
        hooks.append(getattr(enc, "transformer%d" % i).register_forward_hook(
            lambda m, inp, o, i=i: taps.__setitem__("gpt%d_img" % i, o[0].detach().clone())))
    pred = model(*args)
    loss = torch.nn.functional.l1_loss(pred, batch["gt_wp"], reduction="none").mean()  # phase2:104
    loss.backward()
    for h in hooks:
        h.remove()
    res["train_pred_wp"] = pred.detach().numpy()
    res["train_loss"] = np.float32(loss.item())
    for k, v in taps.items():
        v = v.double()
        res["tap_%s_sum" % k] = np.float64(v.sum().item())
        res["tap_%s_abs" % k] = np.float64(v.abs().sum().item())
        res["tap_%s_head" % k] = v.flatten()[:16].float().numpy()
    gnorm, ghead, gnone = [], [], []
    params = dict(model.named_parameters())
    for k in pnames:
        g = params[k].grad
        gnone.append(g is None)
        if g is None:
            gnorm.append(0.0)
            ghead.append(np.zeros(8, np.float32))
        else:
            gnorm.append(float(g.double().norm().item()))
            h8 = np.zeros(8, np.float32)
            flat = g.flatten()[:8].numpy()
            h8[:flat.size] = flat
            ghead.append(h8)
    res["grad_norm"] = np.array(gnorm, np.float64)
    res["grad_head"] = np.stack(ghead)
    res["grad_none"] = np.array(gnone)
    opt.step()
    phead = []
    for k in pnames:
        h8 = np.zeros(8, np.float32)
        flat = params[k].detach().flatten()[:8].numpy()
        h8[:flat.size] = flat
        phead.append(h8)
    res["param_head_after_step"] = np.stack(phead)
    sd = model.state_dict()
    bn_keys = [k for k in sd if k.endswith("running_mean") or k.endswith("running_var")]
    res["bn_keys"] = np.array(bn_keys)
    res["bn_head_after_step"] = np.stack([sd[k].flatten()[:8].numpy().copy() for k in bn_keys])
    res["keys"] = np.array([k for k, _ in keys])
    res["shapes"] = np.array([",".join(map(str, s)) for _, s in keys])
    res["param_names"] = np.array(pnames)
    np.savez_compressed(os.path.join(out_dir, "mmfn_%s_b2.npz" % variant), **res)
    print(variant, "keys", len(keys), "params", sum(p.numel() for p in model.parameters()),
          "eval_loss", res["eval_loss"], "train_loss", res["train_loss"])


def run_frames(ref_models, ref_dl, ref_cfg, out_dir):
    """The image model with more than one frame per sample: (seq_len, n_views) = (2, 1) and (1, 2).  The reference's GPT
    concatenates the n_views*seq_len camera frames, the seq_len LiDAR frames and the seq_len map frames of a sample into
    one token sequence (model_img.py:211-246) and the encoder sums the pooled features of all of them (:410-423).  Frame j
    of every modality comes from fixtures.synthetic_batch(seed=42 + j); labels from seed 42 (harness.frames_args)."""
    res = {}
    for seq_len, n_views in ((2, 1), (1, 2)):
        tag = "s%dv%d_" % (seq_len, n_views)
        torch.manual_seed(0)
        cfg = ref_cfg.GlobalConfig()
        cfg.seq_len, cfg.n_views = seq_len, n_views
        model = ref_models.model_img.MMFN(cfg, "cpu")
        fixtures.fill_module(model)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        batches = [fixtures.synthetic_batch(2, "img", seed=42 + j, n_lidar=16384, lanes=4) for j in range(seq_len * n_views)]
        pre = [reference_inputs(b, ref_dl) for b in batches]
        b0 = batches[0]
        args = ([f for f, _ in pre], [bev for _, bev in pre[:seq_len]], [b["map_u8"].float() for b in batches[:seq_len]],
                [[b0["lane"]], [b0["lane_num"].float()], int(b0["lane_num"].max())], [b0["radar"]], [b0["radar_adj"]],
                b0["target_point"], b0["velocity"])
        gt = batches[0]["gt_wp"]
        res[tag + "pos_emb_shapes"] = np.array([list(getattr(model.encoder, "transformer%d" % i).pos_emb.shape) for i in range(1, 5)])
        bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        for m in bns:
            m.momentum = 1.0
        model.train()
        with torch.no_grad():
            model(*args)
        model.eval()
        with torch.no_grad():
            res[tag + "eval_pred_wp"] = model(*args).numpy()
        for m in bns:
            m.momentum = 0.1
        fixtures.fill_module(model)
        model.train()
        taps = {}
        h = model.encoder.register_forward_hook(lambda m, i, o: taps.__setitem__("fused", o.detach().clone()))
        pred = model(*args)
        loss = torch.nn.functional.l1_loss(pred, gt, reduction="none").mean()
        loss.backward()
        h.remove()
        res[tag + "train_pred_wp"] = pred.detach().numpy()
        res[tag + "train_loss"] = np.float32(loss.item())
        res[tag + "fused"] = taps["fused"].numpy()
        names, gnorm, ghead = [], [], []
        for k, p_ in model.named_parameters():
            names.append(k)
            gnorm.append(float(p_.grad.double().norm().item()))
            h8 = np.zeros(8, np.float32)
            flat = p_.grad.flatten()[:8].numpy()
            h8[:flat.size] = flat
            ghead.append(h8)
        res[tag + "param_names"] = np.array(names)
        res[tag + "grad_norm"] = np.array(gnorm, np.float64)
        res[tag + "grad_head"] = np.stack(ghead)
        print("img frames seq_len %d n_views %d: tokens %d, eval wp[0,0] %s, train_loss %s" % (
            seq_len, n_views, res[tag + "pos_emb_shapes"][0][1], res[tag + "eval_pred_wp"][0, 0], res[tag + "train_loss"]))
    np.savez_compressed(os.path.join(out_dir, "mmfn_img_frames.npz"), **res)


def preprocessing_vectors(ref_dl, ref_du, out_dir):
    res = {}
    # histogram edge cases: bin edges, z == -2.0 boundary, clip > 5, out of range, right-closed last bin
    pts = np.array([
        [-16.0, -24.0, -2.0], [16.0, 8.0, -2.0], [15.999, 7.999, 0.0], [16.001, 0.0, 0.0],
        [0.0, 0.0, -1.9999], [0.0, 0.0, -2.0001], [-16.0001, 0.0, 0.0], [0.124, -0.126, 1.0],
        [1e6, 0.0, 0.0], [0.125, -0.125, 1.0],
    ] + [[3.3, -3.3, 0.5]] * 7 + [[-5.01, 2.2, -2.5]] * 5, dtype=np.float64)
    res["hist_pts"] = pts
    res["hist_out"] = ref_dl.lidar_to_histogram_features(pts, crop=256)
    rng = np.random.RandomState(7)
    big = np.stack([rng.uniform(-20, 20, 4096), rng.uniform(-28, 12, 4096), rng.uniform(-3, 1, 4096)], 1)
    big32 = big.astype(np.float32)
    res["hist_rand_pts"] = big32
    res["hist_rand_out"] = ref_dl.lidar_to_histogram_features(big32.astype(np.float64), crop=256)

    ramp = (np.arange(300 * 400 * 3, dtype=np.int64) % 251).astype(np.uint8).reshape(300, 400, 3)

    class _Img:
        height, width = 300, 400

        def resize(self, size):
            return ramp

    res["crop_out"] = ref_dl.scale_and_crop_image(_Img(), scale=1, crop=256)
    r_small = rng.randn(50, 5)
    r_big = rng.randn(100, 5)
    r_big[:, 3] = np.abs(r_big[:, 3]) + 0.5
    res["radar_small"], res["radar_big"] = r_small, r_big
    res["radar_small_out"] = ref_dl.radar_to_size(r_small, (81, 5))
    res["radar_big_out"] = ref_dl.radar_to_size(r_big, (81, 5))
    # collate structure (data_utils.py:9-67)
    samples = []
    for n in (5, 9, 3):
        samples.append({"vectormaps": [torch.from_numpy(rng.randn(n, 10, 5))],
                        "velocity": float(n), "target_point": (1.0 * n, 2.0 * n),
                        "fronts": [torch.full((3, 4, 4), n, dtype=torch.uint8)]})
    col = ref_du.collate_single_cpu(samples)
    res["collate_lane"] = col["vectormaps"][0][0].numpy()
    res["collate_lane_num"] = col["vectormaps"][0][1].numpy()
    res["collate_lmax"] = np.int64(col["vectormaps"][0][2])
    res["collate_velocity"] = col["velocity"].numpy()
    res["collate_target"] = np.stack([t.numpy() for t in col["target_point"]])
    res["collate_fronts"] = col["fronts"][0].numpy()
    res["collate_src"] = np.concatenate([s["vectormaps"][0].numpy().reshape(-1) for s in samples])
    np.savez_compressed(os.path.join(out_dir, "preprocess.npz"), **res)


def dataio_vectors(ref_dl, ref_du, out_dir):
    """Full-schema collate (data_utils.py:9-67), PRE_Data's radar_adj (dataloader.py:381-384), and the pose /
    command-point geometry of CARLA_Data.__getitem__ (dataloader.py:233-261) evaluated with the reference's
    own transform_2d_points."""
    res = {}
    samples = fixtures.synthetic_samples()
    for s in samples:  # PRE_Data.__getitem__ adds the adjacency row by row
        s["radar_adj"] = np.array([s["radar"][0][:, 1] - s["radar"][0][i, 1] for i in range(81)])
    col = ref_du.collate_single_cpu(samples)
    res["radar_adj"] = col["radar_adj"].numpy()
    res["lane"] = col["vectormaps"][0][0].numpy()
    res["lane_num"] = col["vectormaps"][0][1].numpy()
    res["lmax"] = np.int64(col["vectormaps"][0][2])
    res["radar"] = col["radar"][0].numpy()
    res["waypoints"] = np.stack([np.stack([c.numpy() for c in wp]) for wp in col["waypoints"]])  # [5, 2, B]
    res["target_point"] = np.stack([c.numpy() for c in col["target_point"]])
    for k in ("steer", "throttle", "brake", "command", "velocity"):
        res[k] = col[k].numpy()
    for k in ("fronts", "lidars", "maps"):
        t = col[k][0]
        res[k + "_shape"] = np.array(t.shape)
        res[k + "_dtype"] = np.array(str(t.dtype))
        res[k + "_sum"] = np.float64(t.double().sum().item())
    res["dtypes"] = np.array([str(col[k].dtype) for k in ("steer", "throttle", "brake", "command", "velocity", "radar_adj")]
                             + [str(col["vectormaps"][0][0].dtype), str(col["vectormaps"][0][1].dtype), str(col["radar"][0].dtype),
                                str(col["waypoints"][0][0].dtype), str(col["target_point"][0].dtype)])
    # geometry
    rng = np.random.RandomState(5)
    pts = rng.randn(64, 3) * 10.0
    res["tf_pts"] = pts
    res["tf_args"] = np.array([0.3, 1.5, -2.0, -0.7, 4.0, 0.25])
    res["tf_out"] = ref_dl.transform_2d_points(pts, *res["tf_args"])
    xs, ys, th = rng.randn(5) * 5.0, rng.randn(5) * 5.0, rng.uniform(-3, 3, 5)
    res["pose_x"], res["pose_y"], res["pose_theta"] = xs, ys, th
    ego = 0
    res["pose_waypoints"] = np.array([ref_dl.transform_2d_points(np.zeros((1, 3)), np.pi / 2 - th[i], -xs[i], -ys[i],
                                                                  np.pi / 2 - th[ego], -xs[ego], -ys[ego])[0, :2] for i in range(5)])
    R = np.array([[np.cos(np.pi / 2 + th[ego]), -np.sin(np.pi / 2 + th[ego])],
                  [np.sin(np.pi / 2 + th[ego]), np.cos(np.pi / 2 + th[ego])]])
    cmd = np.array([7.5, -3.25])
    res["pose_cmd"] = cmd
    res["pose_target"] = R.T.dot(np.array([cmd[0] - xs[ego], cmd[1] - ys[ego]]))
    np.savez_compressed(os.path.join(out_dir, "dataio.npz"), **res)


def raw_route_vectors(ref_dl, ref_cfg, out_dir):
    """CARLA_Data (dataloader.py:11-268) run on the synthetic recorded route of fixtures.write_synthetic_route."""
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        fixtures.write_synthetic_route(tmp)
        cfg = ref_cfg.GlobalConfig()
        ds = ref_dl.CARLA_Data([tmp], cfg)
        res["n"] = np.int64(len(ds))
        for i in range(len(ds)):
            s = ds[i]
            # uint8 frames are incompressible noise: keep their SHA-256 (exact comparison) instead of 200 KB each
            res["fronts%d_sha" % i] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(s["fronts"][0].numpy()).tobytes()).digest(), dtype=np.uint8)
            res["fronts%d_shape" % i] = np.array(s["fronts"][0].shape)
            res["lidars%d" % i] = s["lidars"][0]
            res["maps%d_sha" % i] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(s["maps"][0].numpy()).tobytes()).digest(), dtype=np.uint8)
            res["lanes%d" % i] = s["vectormaps"][0].numpy()
            res["radar%d" % i] = s["radar"][0]
            res["waypoints%d" % i] = np.array(s["waypoints"])
            res["target%d" % i] = np.array(s["target_point"])
            res["labels%d" % i] = np.array([s["steer"], s["throttle"], float(s["brake"]), float(s["command"]), s["velocity"]])
        # ---- seq_len = 2.  CARLA_Data's frame loop ends BEFORE the sweep is transformed and appended (dataloader.py:225-232 sit
        # outside `for i in range(self.seq_len)`), so the reference sample carries seq_len camera / map / lane / radar frames and
        # ONE histogram - the last frame's.  Recorded as it is (key prefix s2_), together with what that block computes when it is
        # applied to every frame (s2_lidars_all: the reference's own transform_2d_points + lidar_to_histogram_features, frame by
        # frame): the past sweep moved into the ego frame of the last one.
        cfg2 = ref_cfg.GlobalConfig()
        cfg2.seq_len = 2
        ds2 = ref_dl.CARLA_Data([tmp], cfg2)
        res["s2_n"] = np.int64(len(ds2))
        for i in range(len(ds2)):
            s = ds2[i]
            assert len(s["fronts"]) == 2 and len(s["lidars"]) == 1
            for j in range(2):
                res["s2_fronts%d_%d_sha" % (i, j)] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(s["fronts"][j].numpy()).tobytes()).digest(), dtype=np.uint8)
                res["s2_maps%d_%d_sha" % (i, j)] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(s["maps"][j].numpy()).tobytes()).digest(), dtype=np.uint8)
                res["s2_lanes%d_%d" % (i, j)] = s["vectormaps"][j].numpy()
                res["s2_radar%d_%d" % (i, j)] = s["radar"][j]
            res["s2_lidar_last%d" % i] = s["lidars"][0]
            res["s2_waypoints%d" % i] = np.array(s["waypoints"])
            res["s2_target%d" % i] = np.array(s["target_point"])
            res["s2_labels%d" % i] = np.array([s["steer"], s["throttle"], float(s["brake"]), float(s["command"]), s["velocity"]])
            xs, ys, th = ds2.x[i], ds2.y[i], [0.0 if np.isnan(t) else t for t in ds2.theta[i]]
            per_frame = []
            for j in range(2):
                pts = np.load(ds2.lidar[i][j])[..., :3]
                pts[:, 1] *= -1
                pts = ref_dl.transform_2d_points(pts, np.pi / 2 - th[j], -xs[j], -ys[j], np.pi / 2 - th[1], -xs[1], -ys[1])
                per_frame.append(ref_dl.lidar_to_histogram_features(pts, crop=256))
            res["s2_lidars_all%d" % i] = np.stack(per_frame)
            assert np.array_equal(per_frame[1], s["lidars"][0])
    np.savez_compressed(os.path.join(out_dir, "raw_route.npz"), **res)


def pid_vectors(ref_models, ref_cfg, out_dir):
    cfg = ref_cfg.GlobalConfig()
    # control_pid only touches config + the two PID controllers; avoid building a 105 M-param net
    net = ref_models.model_vec.MMFN.__new__(ref_models.model_vec.MMFN)
    torch.nn.Module.__init__(net)
    net.config = cfg
    P = ref_models.model_vec.PIDController
    net.turn_controller = P(cfg.turn_KP, cfg.turn_KI, cfg.turn_KD, cfg.turn_n)
    net.speed_controller = P(cfg.speed_KP, cfg.speed_KI, cfg.speed_KD, cfg.speed_n)
    rng = np.random.RandomState(3)
    wps = rng.randn(5, 1, 4, 2).astype(np.float32) * 2.0
    vels = np.abs(rng.randn(5, 1)).astype(np.float32) * 3.0
    vels[2] = 0.0
    outs = []
    for wp, v in zip(wps, vels):
        s, t, b, meta = net.control_pid(torch.from_numpy(wp.copy()), torch.from_numpy(v.copy()))
        outs.append([float(s), float(t), float(b), meta["angle"], meta["desired_speed"], meta["delta"]])
    np.savez_compressed(os.path.join(out_dir, "pid.npz"), wps=wps, vels=vels, outs=np.array(outs))


def main():
    if not os.path.isdir(REF):
        print("reference mount %s absent: golden vectors can only be generated in the authoring "
              "container; keeping the committed fixtures" % REF)
        return
    install_shims()
    import importlib
    if "--only-frames" in sys.argv:
        torch.set_num_threads(8)
        run_frames(types.SimpleNamespace(model_img=importlib.import_module("mmfn_utils.models.model_img")),
                   importlib.import_module("mmfn_utils.datasets.dataloader"), importlib.import_module("mmfn_utils.datasets.config"),
                   os.path.join(ROOT, "tests", "golden"))
        return
    ref_models = types.SimpleNamespace(
        model_vec=importlib.import_module("mmfn_utils.models.model_vec"),
        model_img=importlib.import_module("mmfn_utils.models.model_img"),
        model_rad=importlib.import_module("mmfn_utils.models.model_rad"))
    ref_dl = importlib.import_module("mmfn_utils.datasets.dataloader")
    ref_du = importlib.import_module("mmfn_utils.datasets.data_utils")
    ref_cfg = importlib.import_module("mmfn_utils.datasets.config")
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(8)
    preprocessing_vectors(ref_dl, ref_du, out_dir)
    dataio_vectors(ref_dl, ref_du, out_dir)
    raw_route_vectors(ref_dl, ref_cfg, out_dir)
    if "--only-io" in sys.argv:
        return
    pid_vectors(ref_models, ref_cfg, out_dir)
    for variant in ("vec", "img", "rad"):
        run_variant(variant, ref_models, ref_dl, ref_cfg, out_dir)
    run_frames(ref_models, ref_dl, ref_cfg, out_dir)


if __name__ == "__main__":
    main()
