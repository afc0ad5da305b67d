"""Host-side data path (mmfn_amd.data) against vectors produced by the reference's own collate / dataset code."""
import os
import pickle

import numpy as np
import pytest
import torch

from mmfn_amd import data as D
from mmfn_amd.config import GlobalConfig
from oracle import fixtures


@pytest.fixture(scope="module")
def io(golden_dir):
    return np.load(os.path.join(golden_dir, "dataio.npz"))


def _samples():
    samples = fixtures.synthetic_samples()
    for s in samples:
        s["radar_adj"] = D.radar_adjacency(s["radar"][0])
    return samples


def test_collate_matches_reference_layout(io):
    col = D.collate(_samples())
    lane, nums, lmax = col["vectormaps"][0]
    assert np.array_equal(lane.numpy(), io["lane"]) and np.array_equal(nums.numpy(), io["lane_num"]) and lmax == int(io["lmax"])
    assert isinstance(lmax, int)
    assert np.array_equal(col["radar"][0].numpy(), io["radar"])
    assert np.array_equal(col["radar_adj"].numpy(), io["radar_adj"])
    wps = np.stack([np.stack([c.numpy() for c in wp]) for wp in col["waypoints"]])
    assert np.array_equal(wps, io["waypoints"])
    assert np.array_equal(np.stack([c.numpy() for c in col["target_point"]]), io["target_point"])
    for k in ("steer", "throttle", "brake", "command", "velocity"):
        assert np.array_equal(col[k].numpy(), io[k]), k
    for k in ("fronts", "lidars", "maps"):
        t = col[k][0]
        assert list(t.shape) == list(io[k + "_shape"]) and str(t.dtype) == str(io[k + "_dtype"])
        assert t.double().sum().item() == float(io[k + "_sum"])
    got = [str(col[k].dtype) for k in ("steer", "throttle", "brake", "command", "velocity", "radar_adj")] + \
          [str(lane.dtype), str(nums.dtype), str(col["radar"][0].dtype), str(col["waypoints"][0][0].dtype),
           str(col["target_point"][0].dtype)]
    assert got == list(io["dtypes"])


def test_collate_rejects_ragged_sequences():
    a, b = _samples()[:2]
    b = dict(b, waypoints=b["waypoints"][:4])
    with pytest.raises(RuntimeError):
        D.collate([a, b])


def test_radar_and_pose_geometry(io, golden_dir):
    pre = np.load(os.path.join(golden_dir, "preprocess.npz"))
    assert np.array_equal(D.radar_to_size(pre["radar_small"]), pre["radar_small_out"])
    assert np.array_equal(D.radar_to_size(pre["radar_big"]), pre["radar_big_out"])
    assert D.radar_to_size(np.zeros((0, 5))).shape == (81, 5)
    out = D.ego_transform(io["tf_pts"], *io["tf_args"])
    assert np.abs(out - io["tf_out"]).max() <= 1e-12
    wps = np.array(D.local_waypoints(io["pose_x"], io["pose_y"], io["pose_theta"], 0))
    assert np.abs(wps - io["pose_waypoints"]).max() <= 1e-12
    tgt = D.local_target_point(io["pose_cmd"][0], io["pose_cmd"][1], io["pose_x"][0], io["pose_y"][0], io["pose_theta"][0])
    assert np.abs(np.array(tgt) - io["pose_target"]).max() <= 1e-12


def test_frame_store_and_staging(tmp_path, io):
    cfg = GlobalConfig()
    for i, s in enumerate(fixtures.synthetic_samples()):
        with open(tmp_path / ("%d.pkl" % i), "wb") as fd:
            pickle.dump(s, fd)
    (tmp_path / "notes.txt").write_text("ignored")
    store = D.FrameStore(str(tmp_path), cfg, "train")
    assert len(store) == 3
    assert os.path.exists(tmp_path / "rg_vec_mmfn_diag_pl_1_4_train.npy")
    assert len(D.FrameStore(str(tmp_path), cfg, "train")) == 3  # second open reads the cached file list
    loader = D.make_loader(store, batch_size=3, num_workers=0)
    batch = next(iter(loader))
    order = [int(os.path.basename(f).split(".")[0]) for f in store.files]
    assert np.array_equal(batch["radar_adj"].numpy(), io["radar_adj"][order])
    args, gt = D.stage_batch(batch, "cpu", cfg, non_blocking=False)
    fronts, lidars, maps, vm, radar, adj, tp, vel = args
    assert fronts[0].dtype == torch.float32 and fronts[0].shape == (3, 3, 256, 256)
    assert vm[0][0].dtype == torch.float32 and vm[0][0].shape == (3, 9, 10, 5) and vm[2] == 9
    assert vm[1][0].dtype == torch.float32 and sorted(vm[1][0].tolist()) == [3.0, 5.0, 9.0]
    assert radar[0].shape == (3, 81, 5) and adj[0].shape == (3, 81, 81)
    assert tp.shape == (3, 2) and vel.shape == (3,) and gt.shape == (3, 4, 2) and gt.dtype == torch.float32
    # waypoints[1:5] of sample 0 in file order
    s0 = fixtures.synthetic_samples()[order[0]]
    assert np.allclose(gt[0].numpy(), np.array(s0["waypoints"][1:5], dtype=np.float32))
    # the prefetcher hands out the same thing
    (args2, gt2), = list(D.DevicePrefetcher(loader, "cpu", cfg))
    assert torch.equal(gt2, gt) and torch.equal(args2[0][0], fronts[0])


def _same(a, b, path=""):
    """Recursive equality of two collated batches: container types, dtypes, shapes, values."""
    assert type(a) is type(b), (path, type(a), type(b))
    if isinstance(a, dict):
        assert list(a) == list(b), path
        for k in a:
            _same(a[k], b[k], path + "/" + k)
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for j, (x, y) in enumerate(zip(a, b)):
            _same(x, y, "%s[%d]" % (path, j))
    elif isinstance(a, torch.Tensor):
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), (path, a.dtype, b.dtype, a.shape, b.shape)
    else:
        assert a == b, (path, a, b)


def test_packed_frames_give_the_batches_of_the_pickle_store(tmp_path):
    """pack_frames / PackedFrames / PackedLoader (the loader that feeds the step at its resident rate): every batch equals
    collate() over the same FrameStore samples bit for bit - ragged lanes, float64 labels, bool / int / tuple fields - in any
    index order; the loader covers every sample once per epoch, reshuffles per epoch, honours a sampler and drop_last."""
    cfg = GlobalConfig()
    src = tmp_path / "pkl"
    src.mkdir()
    samples = fixtures.synthetic_samples(lane_counts=(5, 9, 3, 7, 1), seed=5, radar_counts=(50, 100, 81, 3, 90))
    for i, s in enumerate(samples):
        with open(src / ("%d.pkl" % i), "wb") as fd:
            pickle.dump(s, fd)
    store = D.FrameStore(str(src), cfg, "train")
    packed = D.PackedFrames(D.pack_frames(store, str(tmp_path / "packed")))
    assert len(packed) == len(store) == 5
    for idx in ([0, 1, 2, 3, 4], [4, 2], [3], [1, 1, 0]):
        _same(packed.batch(idx), D.collate([store[i] for i in idx]))
    with pytest.raises(IndexError):
        packed.batch([5])
    # the loader
    seen = []
    loader = D.PackedLoader(packed, batch_size=2, shuffle=True, seed=3, pin_memory=False)
    assert len(loader) == 3
    e1 = [b["velocity"].tolist() for b in loader]
    e2 = [b["velocity"].tolist() for b in loader]
    flat = lambda e: sorted(v for b in e for v in b)
    assert flat(e1) == flat(e2) == sorted(float(s["velocity"]) for s in (store[i] for i in range(5)))
    assert e1 != e2                                            # a new permutation every epoch
    assert [len(b) for b in e1] == [2, 2, 1]
    assert len(D.PackedLoader(packed, 2, drop_last=True)) == 2 and len(list(D.PackedLoader(packed, 2, drop_last=True, pin_memory=False))) == 2
    sampler = D.shard_sampler(store, rank=1, world=2, shuffle=False)
    got = [b["velocity"].tolist() for b in D.PackedLoader(packed, 2, sampler=sampler, pin_memory=False)]
    want = [float(store[i]["velocity"]) for i in sampler]
    assert [v for b in got for v in b] == want
    # staging and the prefetcher take its batches like the DataLoader's
    b0 = next(iter(D.PackedLoader(packed, 3, pin_memory=False)))
    ref = D.collate([store[i] for i in range(3)])
    (a1, g1), (a2, g2) = D.stage_batch(b0, "cpu", cfg, non_blocking=False), D.stage_batch(ref, "cpu", cfg, non_blocking=False)
    _same(list(a1), list(a2))
    assert torch.equal(g1, g2)
    it = iter(D.PackedLoader(packed, 1, pin_memory=False))     # an abandoned iteration must not leave the producer thread blocked
    next(it)
    it.close()
    # a store whose samples disagree in structure is refused
    bad = dict(samples[0]); bad.pop("steer")
    with pytest.raises(ValueError):
        D.pack_frames([samples[0], bad], str(tmp_path / "bad"))


def test_raw_route_reader_with_two_frames_per_sample(tmp_path, golden_dir):
    """RawFrameStore at seq_len = 2 against the reference's CARLA_Data on the same synthetic route (raw_route.npz, s2_*): camera /
    map / lane / radar frames, labels, waypoints, and the LiDAR sweeps moved into the ego frame of the last one.  The reference
    keeps only the last sweep (its transform block sits outside the frame loop, dataloader.py:225-232): that one must be
    bit-identical; the earlier sweep is pinned against the reference's own transform_2d_points + histogram applied to it."""
    import hashlib
    from oracle import fixtures, preprocess
    from mmfn_amd import data as D
    from mmfn_amd.config import GlobalConfig
    g = np.load(os.path.join(golden_dir, "raw_route.npz"))
    root = str(tmp_path / "routes")
    fixtures.write_synthetic_route(root)
    store = D.RawFrameStore([root], GlobalConfig(seq_len=2))
    assert len(store) == int(g["s2_n"]) == 1
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
    s = store[0]
    assert s["rgb_u8"].shape == (2, 300, 400, 3) and s["lidar_in_ego_frame"] and s["lidar_reference_frame"] == 1
    for j in range(2):
        crop = s["rgb_u8"][j].numpy()[150 - 128:150 + 128, 200 - 128:200 + 128]
        assert np.array_equal(sha(np.transpose(crop, (2, 0, 1))), g["s2_fronts0_%d_sha" % j])
        assert np.array_equal(sha(s["maps"][j].numpy()), g["s2_maps0_%d_sha" % j])
        assert np.array_equal(s["vectormaps"][j].numpy(), g["s2_lanes0_%d" % j])
        assert np.array_equal(s["radar"][j], g["s2_radar0_%d" % j])
        hist = preprocess.lidar_histogram(np.asarray(s["lidar_pts"][j])[:, :3])
        assert np.array_equal(hist, g["s2_lidars_all0"][j]), j
    assert np.array_equal(preprocess.lidar_histogram(np.asarray(s["lidar_pts"][1])[:, :3]), g["s2_lidar_last0"])
    assert np.abs(np.array(s["waypoints"]) - g["s2_waypoints0"]).max() <= 1e-12 and len(s["waypoints"]) == 6
    assert np.abs(np.array(s["target_point"]) - g["s2_target0"]).max() <= 1e-12
    lab = np.array([s["steer"], s["throttle"], float(s["brake"]), float(s["command"]), s["velocity"]])
    assert np.array_equal(lab, g["s2_labels0"])
    # collated: frames and sweeps stacked per sample, padded with far points
    b = D.collate_raw([s, s])
    assert b["rgb_u8"].shape == (2, 2, 300, 400, 3) and b["lidar_pts"].shape[:2] == (2, 2) and b["lidar_in_ego_frame"] is True
    assert len(b["maps"]) == 2 and b["maps"][0].shape == (2, 3, 256, 256) and len(b["vectormaps"]) == 2
    # seq_len = 1 keeps its layout
    s1 = D.RawFrameStore([root], GlobalConfig())[0]
    assert s1["rgb_u8"].shape == (300, 400, 3) and not isinstance(s1["lidar_pts"], list) and "lidar_in_ego_frame" not in s1
