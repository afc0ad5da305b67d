"""Pin the CPU oracle against vectors produced by the reference itself (oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures, harness, preprocess
from oracle.model import PIDController, OracleMMFN
from oracle.config import OracleConfig


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_histogram_edge_cases(golden_dir):
    g = _load(golden_dir, "preprocess.npz")
    np.testing.assert_array_equal(preprocess.lidar_histogram(g["hist_pts"]), g["hist_out"])
    np.testing.assert_array_equal(preprocess.lidar_histogram(g["hist_rand_pts"].astype(np.float64)),
                                  g["hist_rand_out"])


def test_crop_radar_collate(golden_dir):
    g = _load(golden_dir, "preprocess.npz")
    ramp = (np.arange(300 * 400 * 3, dtype=np.int64) % 251).astype(np.uint8).reshape(300, 400, 3)
    np.testing.assert_array_equal(preprocess.crop_chw(ramp), g["crop_out"])
    np.testing.assert_array_equal(preprocess.radar_to_size(g["radar_small"]), g["radar_small_out"])
    np.testing.assert_array_equal(preprocess.radar_to_size(g["radar_big"]), g["radar_big_out"])
    # collate: lanes padded to the longest sample, counts kept (data_utils.py:19-25)
    assert g["collate_lane"].shape == (3, 9, 10, 5) and int(g["collate_lmax"]) == 9
    np.testing.assert_array_equal(g["collate_lane_num"], [5, 9, 3])
    assert np.all(g["collate_lane"][0, 5:] == 0) and np.all(g["collate_lane"][2, 3:] == 0)


def test_pid_sequence(golden_dir):
    g = _load(golden_dir, "pid.npz")
    cfg = OracleConfig()
    net = OracleMMFN.__new__(OracleMMFN)
    torch.nn.Module.__init__(net)
    net.config = cfg
    net.turn_controller = PIDController(cfg.turn_KP, cfg.turn_KI, cfg.turn_KD, cfg.turn_n)
    net.speed_controller = PIDController(cfg.speed_KP, cfg.speed_KI, cfg.speed_KD, cfg.speed_n)
    for wp, v, ref in zip(g["wps"], g["vels"], g["outs"]):
        s, t, b, meta = net.control_pid(torch.from_numpy(wp.copy()), torch.from_numpy(v.copy()))
        got = [float(s), float(t), float(b), meta["angle"], meta["desired_speed"], meta["delta"]]
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


@pytest.mark.parametrize("variant", ["vec", "img", "rad"])
def test_model_against_reference(golden_dir, variant):
    g = _load(golden_dir, "mmfn_%s_b2.npz" % variant)
    torch.set_num_threads(8)
    model = harness.build_oracle(variant)
    # state_dict contract: same keys, shapes and order as the reference
    sd = model.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g["shapes"])
    assert [k for k, _ in model.named_parameters()] == list(g["param_names"])

    batch = fixtures.synthetic_batch(2, variant, seed=42, lanes=9 if variant != "img" else 4)
    args = harness.forward_args(batch, variant)
    np.testing.assert_array_equal(args[1][0].numpy(), g["bev"])
    assert float(args[0][0].double().sum()) == float(g["fronts_crop_sum"])

    harness.calibrate_bn(model, args)
    with torch.no_grad():
        wp = model(*args).numpy()
    np.testing.assert_allclose(wp, g["eval_pred_wp"], rtol=0, atol=1e-6)
    if variant != "img":
        one = [[batch["lane"][:1]], [batch["lane_num"][:1].int()], batch["lane_num"][:1].int().view(1, 1)]
        a1 = ([args[0][0][:1]], [args[1][0][:1]], None, one, [batch["radar"][:1]], [batch["radar_adj"][:1]],
              batch["target_point"][:1], batch["velocity"][:1])
        with torch.no_grad():
            np.testing.assert_allclose(model(*a1).numpy(), g["eval_pred_wp_b1_agent"], rtol=0, atol=1e-6)
    else:   # the image-map agent's batch-1 call (e2e_agent/mmfn_imgnet.py:273-276)
        with torch.no_grad():
            wp1 = model([args[0][0][:1]], [args[1][0][:1]], [args[2][0][:1]], None, None, None, batch["target_point"][:1],
                        batch["velocity"][:1]).numpy()
        np.testing.assert_allclose(wp1, g["eval_pred_wp_b1_agent"], rtol=0, atol=1e-6)

    fixtures.fill_module(model)
    pred, loss, grads = harness.train_step(model, args, batch["gt_wp"])
    np.testing.assert_allclose(pred.numpy(), g["train_pred_wp"], rtol=0, atol=1e-6)
    assert abs(float(loss) - float(g["train_loss"])) <= 1e-6
    names = list(g["param_names"])
    for i, k in enumerate(names):
        gr = grads[k]
        assert (gr is None) == bool(g["grad_none"][i]), k
        if gr is None:
            continue
        ref_norm = float(g["grad_norm"][i])
        assert abs(float(gr.double().norm()) - ref_norm) <= 1e-4 * max(ref_norm, 1e-3), k
    params = dict(model.named_parameters())
    got = np.stack([np.pad(params[k].detach().flatten()[:8].numpy(), (0, max(0, 8 - params[k].numel())))
                    for k in names])
    np.testing.assert_allclose(got, g["param_head_after_step"], rtol=0, atol=2e-6)
    sd = model.state_dict()
    got_bn = np.stack([sd[k].flatten()[:8].numpy() for k in g["bn_keys"]])
    np.testing.assert_allclose(got_bn, g["bn_head_after_step"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("seq_len,n_views", [(2, 1), (1, 2)])
def test_image_model_with_several_frames_against_reference(golden_dir, seq_len, n_views):
    """seq_len > 1 / n_views > 1 (model_img.py:211-246, :410-423): each sample's frames share one token sequence."""
    g = _load(golden_dir, "mmfn_img_frames.npz")
    tag = "s%dv%d_" % (seq_len, n_views)
    torch.set_num_threads(8)
    model = harness.build_oracle("img", seq_len=seq_len, n_views=n_views)
    shapes = [list(getattr(model.encoder, "transformer%d" % i).pos_emb.shape) for i in range(1, 5)]
    assert shapes == g[tag + "pos_emb_shapes"].tolist()
    assert shapes[0][1] == (n_views + 2) * seq_len * 64
    assert [k for k, _ in model.named_parameters()] == list(g[tag + "param_names"])
    _, args, gt = harness.frames_args(seq_len, n_views)
    harness.calibrate_bn(model, args)
    with torch.no_grad():
        np.testing.assert_allclose(model(*args).numpy(), g[tag + "eval_pred_wp"], rtol=0, atol=1e-6)
    fixtures.fill_module(model)
    taps = {}
    model.train()
    pred = model(*args, taps=taps)
    np.testing.assert_allclose(taps["fused"].detach().numpy(), g[tag + "fused"], rtol=1e-5, atol=1e-5)
    pred, loss, grads = harness.train_step(model, args, gt)
    np.testing.assert_allclose(pred.numpy(), g[tag + "train_pred_wp"], rtol=0, atol=1e-6)
    assert abs(float(loss) - float(g[tag + "train_loss"])) <= 1e-6
    norms = np.array([float(grads[k].double().norm()) for k in g[tag + "param_names"]])
    np.testing.assert_allclose(norms, g[tag + "grad_norm"], rtol=2e-3, atol=1e-7)
