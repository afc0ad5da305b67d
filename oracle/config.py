"""Hyper-parameter container restating mmfn_utils/datasets/config.py:3-68 (TEST INFRASTRUCTURE)."""


class OracleConfig(object):
    seq_len = 1
    pred_len = 4
    n_views = 1
    input_resolution = 256
    scale = 1
    crop = 256
    lr = 1e-4
    vert_anchors = 8
    horz_anchors = 8
    n_embd = 512
    block_exp = 4
    n_layer = 8
    n_head = 4
    n_scale = 4
    embd_pdrop = 0.1
    resid_pdrop = 0.1
    attn_pdrop = 0.1
    turn_KP, turn_KI, turn_KD, turn_n = 1.0, 0.65, 0.2, 30
    speed_KP, speed_KI, speed_KD, speed_n = 4.0, 0.4, 0.8, 30
    max_throttle = 0.75
    brake_speed = 0.1
    brake_ratio = 1.1
    clip_delta = 0.25
    hidden = 81
    nb_heads = 2
    alpha = 0.2
    lane_node_num = 10
    feature_num = 5
    lane_channels = 7   # 8 = perf-only pre-vectorised [B, L, 19, 8] polylines (see oracle/model.py _VectornetEncoder)

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
