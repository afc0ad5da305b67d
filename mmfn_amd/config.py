"""Hyper-parameters of MMFN (restates mmfn_utils/datasets/config.py:3-68; same attribute names so a
reference GlobalConfig instance and this one are interchangeable as the `config` ctor argument)."""
import os


class GlobalConfig(object):
    # data
    seq_len = 1
    pred_len = 4
    ignore_sides = True
    ignore_rear = True
    n_views = 1
    input_resolution = 256
    scale = 1
    crop = 256
    lr = 1e-4
    # conv encoder
    vert_anchors = 8
    horz_anchors = 8
    anchors = vert_anchors * horz_anchors
    # GPT encoder
    n_embd = 512
    block_exp = 4
    n_layer = 8
    n_head = 4
    n_scale = 4
    embd_pdrop = 0.1
    resid_pdrop = 0.1
    attn_pdrop = 0.1
    # PID controller
    turn_KP, turn_KI, turn_KD, turn_n = 1.0, 0.65, 0.2, 30
    speed_KP, speed_KI, speed_KD, speed_n = 4.0, 0.4, 0.8, 30
    max_throttle = 0.75
    brake_speed = 0.1
    brake_ratio = 1.1
    clip_delta = 0.25
    # radar GAT
    hidden = 81
    nb_heads = 2
    alpha = 0.2
    # vector map
    lane_node_num = 10
    feature_num = 5
    up = down = left = right = 28
    tmp_town_for_save_opendrive = "/tmp/opendrvie_tmp"
    # Subgraph input width.  7 is the reference (model_vec.py:434: vectors built from [.., 10, 5] lane nodes); 8 selects the
    # PERF-ONLY pre-vectorised polyline input [B, L, 19, 8] named by BASELINE.json's north_star
    lane_channels = 7
    # not in the reference: arithmetic of the Linear / Winograd GEMMs.  "f32" is the parity path; "bf16" rounds their operands
    # to bf16 on the way into the MFMA units (fp32 accumulation, activations and master weights; BASELINE configs[2])
    gemm_dtype = "f32"
    # not in the reference: the bf16 TRAINING MODE proper (BASELINE configs[2]).  "bf16": activations, saved tensors and the
    # per-step weight shadows are bf16 in HBM (ResNet trunks, fusion transformers and the glue between them), accumulation /
    # normalisation statistics / master weights / gradients / optimizer / loss head / VectorNet / the two 7x7 stems are fp32
    # (mmfn_gemm_bf16 + the *_bf16 entry points).  Supersedes gemm_dtype.  vec and img variants.
    act_dtype = "f32"

    def __init__(self, **kwargs):
        self.train_data, self.val_towns = [], []
        for k, v in kwargs.items():
            setattr(self, k, v)

    def data_folder(self, args):
        root_dir = os.path.join(args.absolute_path, args.data_folder)
        self.train_data = [os.path.join(root_dir, town + "_short") for town in args.train_towns]
        self.val_data = [os.path.join(root_dir, town + "_short") for town in args.val_towns]
