"""Host-side contract of the product module (no GPU): checkpoint layout, flat HBM layout, ABI."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from mmfn_amd.config import GlobalConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vec_model():
    from mmfn_amd.model import MMFN
    torch.manual_seed(0)
    return MMFN(GlobalConfig(), "cpu")


def test_state_dict_matches_reference_layout(golden_dir, vec_model):
    g = np.load(os.path.join(golden_dir, "mmfn_vec_b2.npz"))
    sd = vec_model.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g["shapes"])
    assert [k for k, _ in vec_model.named_parameters()] == list(g["param_names"])


@pytest.mark.parametrize("variant,cls", [("img", "MMFNImg"), ("rad", "MMFNRad")])
def test_other_variants_layout(golden_dir, variant, cls):
    import mmfn_amd.model as M
    m = getattr(M, cls)(GlobalConfig(), "cpu")
    g = np.load(os.path.join(golden_dir, "mmfn_%s_b2.npz" % variant))
    assert list(m.state_dict().keys()) == list(g["keys"])
    assert [",".join(map(str, v.shape)) for v in m.state_dict().values()] == list(g["shapes"])


def test_checkpoint_roundtrip_with_oracle(vec_model):
    """A reference-layout checkpoint loads strictly; conv weights land in [Cout,KH,KW,Cin] storage."""
    from oracle import harness
    oracle = harness.build_oracle("vec")
    missing = vec_model.load_state_dict(oracle.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    sd, ref = vec_model.state_dict(), oracle.state_dict()
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k
    L = vec_model._layout
    name = "encoder.image_encoder.features.layer2.0.conv1.weight"
    w = ref[name]
    assert torch.equal(L.w(name), w.permute(0, 2, 3, 1))
    assert L.w(name).is_contiguous() and tuple(sd[name].shape) == tuple(w.shape)
    # k/q/v of one block are adjacent: one [3C, C] GEMM operand
    base = "encoder.transformer3.blocks.5.attn."
    packed, _ = L.packed(base + "key.weight", 3 * 256, 256)
    assert torch.equal(packed, torch.cat([ref[base + "key.weight"], ref[base + "query.weight"], ref[base + "value.weight"]]))
    pb, _ = L.packed(base + "key.bias", 3 * 256)
    assert torch.equal(pb, torch.cat([ref[base + "key.bias"], ref[base + "query.bias"], ref[base + "value.bias"]]))
    # the 21 never-trained tensors sit behind the optimizer range
    unused = [n for n in L.names if n in L.unused]
    assert len(unused) == 21 and all(L.offsets[n][0] >= L.tail for n in unused)
    assert all(L.offsets[n][0] + L.offsets[n][1] <= L.tail for n in L.names if n not in L.unused)
    # torch.save / torch.load round trip of the state_dict (what Engine.save does, phase2:207-211)
    import io
    buf = io.BytesIO()
    torch.save(vec_model.state_dict(), buf)
    buf.seek(0)
    back = torch.load(buf)
    for k in ref:
        assert torch.equal(back[k], ref[k]), k


def test_entry_points_resolve():
    """load_entry_point("mmfn_utils.models.model_vec:MMFN") (run_steps/utils.py:68-72)."""
    from importlib import import_module
    for mod, variant in (("model_vec", "vec"), ("model_img", "img"), ("model_rad", "rad")):
        cls = getattr(import_module("mmfn_utils.models." + mod), "MMFN")
        assert cls.variant == variant
    from mmfn_utils.datasets.config import GlobalConfig as G2
    assert G2 is GlobalConfig


def test_forward_on_cpu_fails_loudly(vec_model):
    from mmfn_amd._lib import MMFNLibraryError
    x = torch.zeros(1, 3, 256, 256)
    vm = [[torch.zeros(1, 2, 10, 5)], [torch.tensor([2.0])], 2]
    with pytest.raises(MMFNLibraryError):
        vec_model([x], [torch.zeros(1, 2, 256, 256)], None, vm, None, None, torch.zeros(1, 2), torch.zeros(1))


def test_control_pid_matches_reference(golden_dir, vec_model):
    from mmfn_amd.model import PIDController
    g = np.load(os.path.join(golden_dir, "pid.npz"))
    cfg = vec_model.config
    vec_model.turn_controller = PIDController(cfg.turn_KP, cfg.turn_KI, cfg.turn_KD, cfg.turn_n)
    vec_model.speed_controller = PIDController(cfg.speed_KP, cfg.speed_KI, cfg.speed_KD, cfg.speed_n)
    for wp, v, ref in zip(g["wps"], g["vels"], g["outs"]):
        s, t, b, meta = vec_model.control_pid(torch.from_numpy(wp.copy()), torch.from_numpy(v.copy()))
        got = [float(s), float(t), float(b), meta["angle"], meta["desired_speed"], meta["delta"]]
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


def test_abi_library_exports_every_declared_symbol():
    from mmfn_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mmfn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mmfn_\w+)\s*\(", hdr)))
    assert len(declared) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert [d for d in declared if d not in exported] == []
    handle = _lib.lib()
    assert handle.mmfn_abi_version() == 1
    assert set(_lib._SIGNATURES) == set(declared)


def test_no_hot_kernel_uses_scratch_memory():
    """The build gate (mmfn_amd/build.check_scratch): every kernel hipcc compiled reports 0 bytes of scratch per lane unless
    csrc/scratch_allowlist.txt names it; the fusion transformers' attention kernels (head size 128, T = 192) are never listed."""
    from mmfn_amd import build
    res = build.kernel_resources()
    if not res:
        pytest.skip("no compiler resource remarks beside the objects (library built by an older build.py)")
    assert len(res) >= 400
    build.check_scratch(verbose=False)   # raises on an unlisted kernel with scratch
    hot = [k for k in res if "attn_wg_" in k and "ILi128ELi3E" in k]
    assert len(hot) == 6 and all(res[k]["scratch"] == 0 for k in hot)
    allow = [l.split("#")[0].strip() for l in open(build.SCRATCH_ALLOW)]
    assert [a for a in allow if a] == [], "round 6 emptied the allow-list: a new entry needs a reason a reviewer accepts"
    assert all(v.get("scratch", 0) == 0 for v in res.values())


def test_decay_groups_follow_the_reference_rule():
    """configure_optimizers == GPT.configure_optimizers (model_vec.py:179-209) on the GPT sub-modules, extended to the whole
    model: Linear / Conv2d weights decay; biases, LayerNorm / BatchNorm weights and pos_emb do not."""
    import torch.nn as nn
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFNRad
    from mmfn_amd.optim import FusedAdamW, configure_optimizers
    m = MMFNRad(GlobalConfig(), "cpu")
    groups = configure_optimizers(m)
    name_of = {id(p): n for n, p in m.named_parameters()}
    decay = {name_of[id(p)] for p in groups[0]["params"]}
    no_decay = {name_of[id(p)] for p in groups[1]["params"]}
    assert groups[0]["weight_decay"] == 0.01 and groups[1]["weight_decay"] == 0.0
    assert decay | no_decay == set(name_of.values()) and not decay & no_decay
    # the reference's rule, restated per module type
    for mn, mod in m.named_modules():
        for pn, _ in mod.named_parameters(recurse=False):
            full = "%s.%s" % (mn, pn) if mn else pn
            if pn.endswith("bias") or isinstance(mod, (nn.LayerNorm, nn.BatchNorm2d)) or pn == "pos_emb":
                assert full in no_decay, full
            elif pn.endswith("weight") and isinstance(mod, (nn.Linear, nn.Conv2d)):
                assert full in decay, full
    assert "encoder.transformer1.pos_emb" in no_decay and "decoder.bias_ih" in no_decay and "decoder.weight_hh" in decay
    assert "encoder.radar_encoder.attention_0.W" in decay
    opt = FusedAdamW(m, lr=1e-4, param_groups=groups)   # group-id map builds without a GPU
    gid = opt._group_of
    L = m._layout
    off, n = L.offsets["encoder.transformer1.pos_emb"]
    assert gid.numel() == L.total // 4 and int(gid[off // 4]) == 1 and int(gid[(off + n - 1) // 4]) == 1
    off, n = L.offsets["encoder.transformer4.blocks.0.mlp.0.weight"]
    assert int(gid[off // 4]) == 0
    import pytest
    with pytest.raises(ValueError):
        FusedAdamW(m, param_groups=[{"params": groups[0]["params"]}])   # must cover every parameter


def test_comm_library_exports_every_declared_symbol():
    """libmmfn_comm.so (RCCL gradient all-reduce behind a C ABI, include/mmfn_comm.h): loads and exports every declared entry
    point; no collective is called without a GPU."""
    import ctypes
    import re
    from mmfn_amd import comm
    text = re.sub(r"/\*.*?\*/", " ", open(comm.HEADER_PATH).read(), flags=re.S)
    declared = re.findall(r"\bint\s+(mmfn_\w+)\s*\(", text)
    assert set(declared) == {"mmfn_comm_abi_version", "mmfn_comm_unique_id", "mmfn_comm_init", "mmfn_comm_destroy", "mmfn_comm_ranks",
                             "mmfn_allreduce_sum_f32", "mmfn_allreduce_sum_bf16", "mmfn_broadcast_bytes"}
    handle = comm.lib()
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.mmfn_comm_abi_version() == 1
    assert handle.mmfn_allreduce_sum_f32(None, None, 0, None) == -1     # argument checking happens before any RCCL call
    assert handle.mmfn_allreduce_sum_bf16(None, None, 0, None) == -1


def test_checkpoint_stamps_written_before_the_content_hash_still_compare_equal():
    """ADVICE r4 (trainer.py:239): a recent.log from before the stamps became [size, sha256] holds [size, mtime_ns]; such a
    healthy directory must not look like an interrupted save."""
    from mmfn_amd.trainer import _same_save
    assert _same_save([1234, 1727500000000000000], [1234, "ab" * 32])          # legacy stamp: size decides
    assert not _same_save([1234, 1727500000000000000], [1235, "ab" * 32])
    assert _same_save([1234, "ab" * 32], [1234, "ab" * 32])
    assert not _same_save([1234, "ab" * 32], [1234, "cd" * 32])                # same size, other content
