"""Data side of the training path: the PRE_Data sample store, batch collation and host->device staging.

Mirrors, for the formats either side of the hot path:
  FrameStore        <- PRE_Data                    (mmfn_utils/datasets/dataloader.py:349-385)
  collate           <- collate_single_cpu          (mmfn_utils/datasets/data_utils.py:9-67)
  stage_batch       <- the H2D block of Engine.train (run_steps/phase2_train_net.py:63-103)
  radar_to_size / radar_adjacency / ego_transform / local_waypoints / local_target_point
                    <- dataloader.py:336-346, 381-384, 311-334, 240-261
The sample dict, the collated dict and the tensors handed to MMFN.forward keep the reference's keys,
nesting, dtypes and shapes.  What differs is where bytes are widened: uint8 camera / raster-map frames
cross PCIe as uint8 (the reference widens them to fp32 on the host first, 4x the traffic) and the
float64 lane / radar / label tensors are narrowed to fp32 on the host before the copy.

Host-side only (numpy / torch CPU tensors + async copies): no kernels live here, and nothing here is
part of the CPU oracle.
"""
import os
import pickle

import numpy as np
import torch

RADAR_ROWS = 81


# ------------------------------------------------------------------------------------------ per-sample geometry
def radar_to_size(points, rows=RADAR_ROWS, cols=5):
    """Force a radar return list to exactly `rows` rows (dataloader.py:336-346): short lists are zero
    padded; long ones lose the surplus rows with the largest |depth / velocity| (time to collision),
    in the order a descending argsort names them."""
    pts = np.asarray(points)
    n = pts.shape[0]
    if n < rows:
        out = np.zeros((rows, cols))
        out[:n, :] = pts[:n, :]
        return out
    ttc = np.abs(pts[:, 0] / pts[:, 3])
    surplus = np.argsort(-ttc)[:n - rows]
    return np.delete(pts, surplus, 0)


def radar_adjacency(radar):
    """adj[i, j] = radar[j, 1] - radar[i, 1] (dataloader.py:381-384); only its sign is used downstream."""
    col = np.asarray(radar)[:, 1]
    return col[None, :] - col[:, None]


def ego_transform(xyz, r1, t1_x, t1_y, r2, t2_x, t2_y):
    """Move 2-D points from frame 1 (rotation r1, translation t1) into frame 2 (dataloader.py:311-334).
    The z column rides along unchanged."""
    xyz = np.asarray(xyz, dtype=np.float64)
    h = np.stack([xyz[:, 0], xyz[:, 1], np.ones(len(xyz))], 0)

    def frame(r, tx, ty):
        c, s = np.cos(r), np.sin(r)
        return np.array([[c, s, tx], [-s, c, ty], [0.0, 0.0, 1.0]])

    world = frame(r1, t1_x, t1_y) @ h
    local = np.linalg.inv(frame(r2, t2_x, t2_y)) @ world
    out = local.T.copy()
    out[:, 2] = xyz[:, 2]
    return out


def local_waypoints(xs, ys, thetas, ego_index):
    """Ego-frame positions of the recorded poses (dataloader.py:240-248): pose i's origin seen from the
    pose at `ego_index`, using 90deg - theta as the reference does."""
    ex, ey, et = xs[ego_index], ys[ego_index], thetas[ego_index]
    out = []
    for x, y, t in zip(xs, ys, thetas):
        p = ego_transform(np.zeros((1, 3)), np.pi / 2 - t, -x, -y, np.pi / 2 - et, -ex, -ey)
        out.append((float(p[0, 0]), float(p[0, 1])))
    return out


def local_target_point(x_command, y_command, ego_x, ego_y, ego_theta):
    """Route command point in the ego frame (dataloader.py:250-261): R(90deg + theta)^T (p - ego)."""
    a = np.pi / 2 + ego_theta
    rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    return tuple(rot.T.dot(np.array([x_command - ego_x, y_command - ego_y])))


# ------------------------------------------------------------------------------------------ host cores
def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box may report 256 logical
    CPUs and grant a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def limit_host_threads():
    """torch's intra-op pool defaults to one thread per LOGICAL cpu.  Under a smaller cgroup quota every tiny host-side tensor
    op of the loader / staging code (a float64 -> float32 cast of the lane tensor, a stack of labels) then pays an OpenMP barrier
    across threads that are mostly descheduled - milliseconds each, ~100 ms per step measured on a 256-cpu box with a 16-cpu
    quota (tools/trainer_bench.py: 214 samples/s instead of > 900).  Lowers the pool to the usable cores; never raises it."""
    n = usable_cores()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


# ------------------------------------------------------------------------------------------ sample store
class FrameStore(torch.utils.data.Dataset):
    """Reader of the phase-1 output: one pickle per frame (written by run_steps/phase1_preprocess_data.py:42-48)
    plus the `rg_vec_mmfn_diag_pl_<seq>_<pred>_<use>.npy` file-list cache, created on first use exactly as
    PRE_Data does (dataloader.py:356-371) so both implementations can share a directory."""

    def __init__(self, root, config, data_use="train"):
        self.seq_len, self.pred_len = config.seq_len, config.pred_len
        index = os.path.join(root, "rg_vec_mmfn_diag_pl_%d_%d_%s.npy" % (self.seq_len, self.pred_len, data_use))
        if not os.path.exists(index):
            files = [str(root) + "/" + f for f in os.listdir(root) if f.split(".")[-1] == "pkl"]
            np.save(index, files)
        self.files = np.load(index)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        with open(self.files[i], "rb") as fd:
            sample = pickle.load(fd)
        sample["radar_adj"] = radar_adjacency(sample["radar"][0])
        return sample


PRE_Data = FrameStore  # reference name


# ------------------------------------------------------------------------------------------ packed frames (row f1 at full rate)
# FrameStore unpickles ~0.9 MB per sample on the host; at the step rate of one MI355X (1000 samples/s) that, plus the
# worker -> main-process hand-over of every batch, is the bottleneck of the training loop.  pack_frames
# converts a directory of phase-1 pickles ONCE into flat per-field arrays; PackedFrames memory-maps them and PackedLoader gathers
# a batch's rows straight into pinned staging tensors in a background thread: one memcpy per field, no pickling, no worker
# processes.  The batches are bit-identical to collate([FrameStore[i] for i in indices]) (tests/test_data_cpu.py).
PACK_FORMAT = "mmfn-packed-frames-1"


def _leaf_kind(v):
    if isinstance(v, (torch.Tensor, np.ndarray)):
        return "array"
    if isinstance(v, (float, np.floating)):
        return "float"
    if isinstance(v, (bool, np.bool_)):
        return "bool"
    if isinstance(v, (int, np.integer)):
        return "int"
    if isinstance(v, (str, bytes)):
        return "str"
    if isinstance(v, (tuple, list)):
        return "list"
    raise TypeError("pack_frames: unsupported field type %s" % type(v))


def pack_frames(store, out_dir):
    """FrameStore (or any dataset of phase-1 sample dicts) -> out_dir/{schema.json, <field>.bin ...}.  One streaming pass.
    Every field must have the same structure in all samples (what `collate` requires too); lane sets may be ragged."""
    import json
    os.makedirs(out_dir, exist_ok=True)
    files, scalars, schema = {}, {}, None

    def emit(name, arr):
        arr = np.ascontiguousarray(arr.numpy() if isinstance(arr, torch.Tensor) else np.asarray(arr))
        ent = files.get(name)
        if ent is None:
            ent = files[name] = {"fd": open(os.path.join(out_dir, name + ".bin"), "wb"), "dtype": arr.dtype.str, "shape": list(arr.shape)}
        if arr.dtype.str != ent["dtype"] or list(arr.shape) != ent["shape"]:
            raise ValueError("pack_frames: field %s changes dtype / shape between samples (%s %s vs %s %s)"
                             % (name, arr.dtype.str, list(arr.shape), ent["dtype"], ent["shape"]))
        ent["fd"].write(arr.tobytes())

    def walk(v, name):
        kind = _leaf_kind(v)
        if kind == "array":
            emit(name, v)
            return {"kind": "array", "name": name}
        if kind == "list":
            return {"kind": "list", "items": [walk(x, "%s.%d" % (name, j)) for j, x in enumerate(v)]}
        scalars.setdefault(name, []).append(v.decode() if isinstance(v, bytes) else (v.item() if isinstance(v, np.generic) else v))
        return {"kind": kind, "name": name}

    def lanes(v, name):   # per frame of the sequence: ragged [L, n, F] -> concatenated rows + a count
        items = []
        for j, x in enumerate(v):
            x = np.ascontiguousarray(x.numpy() if isinstance(x, torch.Tensor) else np.asarray(x))
            nm = "%s.%d" % (name, j)
            ent = files.get(nm)
            if ent is None:
                ent = files[nm] = {"fd": open(os.path.join(out_dir, nm + ".bin"), "wb"), "dtype": x.dtype.str, "shape": list(x.shape[1:]), "ragged": True}
            if x.dtype.str != ent["dtype"] or list(x.shape[1:]) != ent["shape"]:
                raise ValueError("pack_frames: lane field %s changes dtype / row shape between samples" % nm)
            ent["fd"].write(x.tobytes())
            scalars.setdefault(nm + ".count", []).append(int(x.shape[0]))
            items.append({"kind": "lanes", "name": nm})
        return {"kind": "list", "items": items}

    n = len(store)
    for i in range(n):
        sample = store[i]
        node = {k: (lanes(v, k) if k == "vectormaps" else walk(v, k)) for k, v in sample.items()}
        if schema is None:
            schema = node
        elif node != schema:
            raise ValueError("pack_frames: sample %d has a different structure than sample 0" % i)
    for ent in files.values():
        ent.pop("fd").close()
    with open(os.path.join(out_dir, "schema.json"), "w") as f:
        json.dump({"format": PACK_FORMAT, "n": n, "fields": schema, "files": files, "scalars": scalars}, f)
    return out_dir


class PackedFrames(object):
    """Memory-mapped view of a pack_frames directory.  batch(indices) == collate([store[i] for i in indices]), bit for bit
    (values, dtypes, shapes, container types), with the tensors optionally in pinned memory."""

    def __init__(self, root):
        import json
        with open(os.path.join(root, "schema.json")) as f:
            meta = json.load(f)
        if meta.get("format") != PACK_FORMAT:
            raise ValueError("%s is not a %s directory" % (root, PACK_FORMAT))
        self.n, self.fields, self.scalars = meta["n"], meta["fields"], meta["scalars"]
        self.arrays, self.offsets = {}, {}
        for name, ent in meta["files"].items():
            dt, shape = np.dtype(ent["dtype"]), tuple(ent["shape"])
            path = os.path.join(root, name + ".bin")
            if ent.get("ragged"):
                counts = np.asarray(self.scalars[name + ".count"], dtype=np.int64)
                self.offsets[name] = np.concatenate([[0], np.cumsum(counts)])
                rows = int(self.offsets[name][-1])
            else:
                rows = self.n
            self.arrays[name] = np.memmap(path, dtype=dt, mode="r", shape=(rows,) + shape) if rows else np.zeros((0,) + shape, dt)

    def __len__(self):
        return self.n

    def _gather(self, node, idx, pin):
        kind = node["kind"]
        if kind == "list":
            return [self._gather(c, idx, pin) for c in node["items"]]
        name = node["name"]
        if kind == "array":
            src = self.arrays[name]
            out = torch.empty((len(idx),) + src.shape[1:], dtype=_TORCH_DTYPE[src.dtype.str], pin_memory=pin)
            np.take(src, idx, axis=0, out=out.numpy(), mode="clip")   # (indices are in range; "clip" writes into `out` directly)
            return out
        if kind == "lanes":   # as _pad_lanes: [padded [B, Lmax, ...], counts i64 [B], int Lmax]
            src, off = self.arrays[name], self.offsets[name]
            nums = torch.tensor([int(off[i + 1] - off[i]) for i in idx])
            lmax = int(nums.max().item())
            out = torch.zeros((len(idx), lmax) + src.shape[1:], dtype=_TORCH_DTYPE[src.dtype.str], pin_memory=pin)
            o = out.numpy()
            for b, i in enumerate(idx):
                o[b, :off[i + 1] - off[i]] = src[off[i]:off[i + 1]]
            return [out, nums, lmax]
        vals = self.scalars[name]
        if kind == "float":
            return torch.tensor([float(vals[i]) for i in idx], dtype=torch.float64)
        if kind in ("int", "bool"):
            return torch.tensor([vals[i] for i in idx])
        if kind == "str":
            return [vals[i] for i in idx]
        raise ValueError(kind)

    def batch(self, indices, pin=False):
        idx = np.asarray(list(indices), dtype=np.int64)
        if idx.size and (idx.min() < 0 or idx.max() >= self.n):
            raise IndexError("sample index out of range")
        return {k: self._gather(node, idx, pin) for k, node in self.fields.items()}

    def __getitem__(self, i):
        """One sample as a batch of one (debugging aid; training goes through PackedLoader)."""
        return self.batch([i])


_TORCH_DTYPE = {np.dtype(k).str: v for k, v in (("uint8", torch.uint8), ("int8", torch.int8), ("int16", torch.int16), ("int32", torch.int32),
                                                  ("int64", torch.int64), ("float16", torch.float16), ("float32", torch.float32),
                                                  ("float64", torch.float64), ("bool", torch.bool))}


class PackedLoader(object):
    """Iterates collated batches of a PackedFrames store; a background thread assembles them `prefetch` batches ahead into
    pinned memory (the caching host allocator keeps a block until the asynchronous copy that reads it has completed).
    Same role as make_loader(FrameStore, ...) for Trainer.train / DevicePrefetcher; `sampler` takes a DistributedSampler."""

    def __init__(self, packed, batch_size, shuffle=False, sampler=None, seed=0, drop_last=False, prefetch=3, pin_memory=True):
        self.packed, self.batch_size, self.shuffle, self.sampler = packed, int(batch_size), shuffle, sampler
        self.seed, self.drop_last, self.prefetch = seed, drop_last, max(1, int(prefetch))
        self.pin = bool(pin_memory) and torch.cuda.is_available()
        self.epoch = 0

    def _order(self):
        if self.sampler is not None:
            return [int(i) for i in self.sampler]
        n = len(self.packed)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            return torch.randperm(n, generator=g).tolist()
        return list(range(n))

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.packed)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        import queue
        import threading
        order = self._order()
        self.epoch += 1
        B = self.batch_size
        chunks = [order[i:i + B] for i in range(0, len(order), B)]
        if self.drop_last and chunks and len(chunks[-1]) < B:
            chunks.pop()
        q, stop = queue.Queue(maxsize=self.prefetch), threading.Event()

        def work():
            try:
                for c in chunks:
                    if stop.is_set():
                        return
                    q.put(self.packed.batch(c, pin=self.pin))
                q.put(None)
            except BaseException as exc:   # hand the failure to the consumer instead of dying silently
                q.put(exc)

        t = threading.Thread(target=work, name="mmfn-packed-loader", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            while t.is_alive():    # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(timeout=0.05)


# ------------------------------------------------------------------------------------------ collation
def _stack(items):
    first = items[0]
    if isinstance(first, torch.Tensor):
        return torch.stack(list(items), 0)
    if isinstance(first, np.ndarray):
        return torch.stack([torch.as_tensor(x) for x in items], 0)
    if isinstance(first, (float, np.floating)):
        return torch.tensor([float(x) for x in items], dtype=torch.float64)
    if isinstance(first, (bool, int, np.integer)):
        return torch.tensor(list(items))
    if isinstance(first, (str, bytes)):
        return list(items)
    if isinstance(first, (tuple, list)):
        width = len(first)
        if any(len(x) != width for x in items):
            raise RuntimeError("each element in list of batch should be of equal size")
        return [_stack([x[j] for x in items]) for j in range(width)]
    raise TypeError("collate: unsupported field type %s" % type(first))


def _pad_lanes(lanes):
    """Ragged per-sample lane sets [L_b, 10, 5] -> [padded [B, Lmax, 10, 5], lane_nums i64[B], int Lmax]
    (data_utils.py:19-25)."""
    lanes = [torch.as_tensor(x) for x in lanes]
    nums = torch.tensor([x.shape[0] for x in lanes])
    lmax = int(nums.max().item())
    out = lanes[0].new_zeros((len(lanes), lmax) + tuple(lanes[0].shape[1:]))
    for b, x in enumerate(lanes):
        out[b, :x.shape[0]] = x
    return [out, nums, lmax]


def collate(samples):
    """List of FrameStore samples -> the batch dict Engine.train consumes."""
    out = {}
    for key in samples[0]:
        column = [s[key] for s in samples]
        if key == "vectormaps":
            seq = len(column[0])
            out[key] = [_pad_lanes([c[i] for c in column]) for i in range(seq)]
        else:
            out[key] = _stack(column)
    return out


collate_single_cpu = collate  # reference name


# ------------------------------------------------------------------------------------------ host -> device
def stage_batch(data, device, config, non_blocking=True):
    """Collated batch -> (MMFN.forward argument tuple, gt_waypoints [B, pred_len, 2]) on `device`
    (phase2_train_net.py:63-103).  uint8 frames are copied as uint8 and widened on the device."""
    dev = torch.device(device)

    def f32(t, narrow_first=True):
        t = torch.as_tensor(t)
        if t.dtype == torch.uint8:  # 1 byte/pixel over PCIe, widened by the GPU
            return t.to(dev, non_blocking=non_blocking).to(torch.float32)
        if narrow_first and t.dtype != torch.float32:
            t = t.to(torch.float32)
        return t.to(dev, non_blocking=non_blocking)

    n = config.seq_len
    fronts = [f32(data["fronts"][i]) for i in range(n)]
    lidars = [f32(data["lidars"][i]) for i in range(n)]
    maps = [f32(data["maps"][i]) for i in range(n)]
    lanes = [f32(data["vectormaps"][i][0]) for i in range(n)]
    lane_nums = [f32(data["vectormaps"][i][1]) for i in range(n)]
    vectormaps = [lanes, lane_nums, data["vectormaps"][0][2]]
    radar = [f32(data["radar"][i]) for i in range(n)]
    radar_adj = [f32(data["radar_adj"]) for _ in range(n)]
    velocity = f32(data["velocity"])
    target_point = f32(torch.stack(list(data["target_point"]), dim=1))
    wps = data["waypoints"]
    gt = torch.stack([torch.stack(list(wps[i]), dim=1) for i in range(n, len(wps))], dim=1)
    return (fronts, lidars, maps, vectormaps, radar, radar_adj, target_point, velocity), f32(gt)


class DevicePrefetcher(object):
    """Iterates a loader of collated batches one batch ahead: the next batch's host->device copies run on a
    dedicated copy stream while the current step computes (the reference copies synchronously inside the
    step, phase2_train_net.py:78-91)."""

    def __init__(self, loader, device, config, variant="vec"):
        limit_host_threads()
        self.loader, self.device, self.config, self.variant = loader, torch.device(device), config, variant
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    def _do_stage(self, data, non_blocking):
        if "rgb_u8" in data:  # RawFrameStore / collate_raw batches
            return stage_raw_batch(data, self.device, self.config, self.variant, non_blocking=non_blocking)
        return stage_batch(data, self.device, self.config, non_blocking=non_blocking)

    def _stage(self, data):
        if self.stream is None:
            return self._do_stage(data, False), None
        with torch.cuda.stream(self.stream):
            staged = self._do_stage(_pin(data), True)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return staged, ready

    def __iter__(self):
        nxt = None
        for data in self.loader:
            cur, nxt = nxt, self._stage(data)
            if cur is not None:
                yield self._hand_over(cur)
        if nxt is not None:
            yield self._hand_over(nxt)

    def _hand_over(self, item):
        staged, ready = item
        if ready is not None:
            torch.cuda.current_stream(self.device).wait_event(ready)
            for t in _tensors(staged):
                t.record_stream(torch.cuda.current_stream(self.device))
        return staged


def _pin(obj):
    if isinstance(obj, torch.Tensor):
        return obj if obj.is_pinned() else obj.pin_memory()
    if isinstance(obj, dict):
        return {k: _pin(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_pin(v) for v in obj)
    return obj


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            for t in _tensors(v):
                yield t
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            for t in _tensors(v):
                yield t


def make_loader(store, batch_size, shuffle=False, sampler=None, num_workers=8, pin_memory=False):
    """DataLoader with this module's collate (phase2_train_net.py:268-274); raw-route stores get collate_raw."""
    fn = collate_raw if isinstance(store, RawFrameStore) else collate
    return torch.utils.data.DataLoader(store, batch_size=batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                                       num_workers=num_workers, pin_memory=pin_memory, collate_fn=fn)


def shard_sampler(store, rank, world, shuffle=True, seed=0):
    """Per-rank sample shard, as DistributedSampler does for the reference (phase2_train_net.py:265-266)."""
    return torch.utils.data.distributed.DistributedSampler(store, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed)


# ------------------------------------------------------------------------------------------ raw recorded routes (row f2)
class RawFrameStore(torch.utils.data.Dataset):
    """Reader of the recorded CARLA routes themselves (`CARLA_Data`, dataloader.py:11-268): <root>/<route>/{rgb_front,
    maps,lidar,radar,vectormap,measurements}/NNNN.{png,npy,json}.  Frame indexing follows the reference exactly
    (first frame of a route skipped, the last pred_len + 1 frames have no future waypoints, dataloader.py:72-75).

    Unlike the reference, a sample carries the RAW sensor frames - the uint8 camera image as recorded and the XYZI
    point list - because crop / normalise / y-flip / histogram run on the GPU inside the network's ingest kernels
    (csrc/ingest.hip); the host only decodes files and does the O(10) pose arithmetic.

    seq_len > 1: a sample carries seq_len camera / map / lane / radar frames, and seq_len LiDAR sweeps, each moved into the ego
    frame of the LAST one (y flip + `ego_transform` = the reference's transform_2d_points, dataloader.py:225-231, 311-334) on the
    host in float64 - `lidar_pts` is then a list and `lidar_in_ego_frame` tells the device side not to flip again.  The reference
    itself runs that block once, after its frame loop, and so returns only the last sweep's histogram (its seq_len > 1 samples
    do not fit its own model); `lidar_reference_frame` = seq_len - 1 names that one."""

    def __init__(self, roots, config):
        self.seq_len, self.pred_len = config.seq_len, config.pred_len
        self.frames = []
        for sub_root in ([roots] if isinstance(roots, str) else list(roots)):
            for route in sorted(os.listdir(sub_root)):
                route_dir = os.path.join(sub_root, route)
                if os.path.isfile(route_dir):
                    continue
                n = (len(os.listdir(os.path.join(route_dir, "rgb_front"))) - self.pred_len - 2) // self.seq_len
                for seq in range(n):
                    ids = ["%04d" % (seq * self.seq_len + 1 + i) for i in range(self.seq_len + self.pred_len)]
                    meas = [_read_json(os.path.join(route_dir, "measurements", k + ".json")) for k in ids]
                    cur = meas[self.seq_len - 1]
                    thetas = [m["theta"] for m in meas]
                    # (the reference zeroes a NaN heading only for the future frames at scan time and for the current
                    # ones at load time, dataloader.py:133-137,224-226: same effect)
                    thetas = [0.0 if np.isnan(t) else t for t in thetas]
                    now = ids[:self.seq_len]
                    self.frames.append({
                        "rgb": [os.path.join(route_dir, "rgb_front", k + ".png") for k in now],
                        "map": [os.path.join(route_dir, "maps", k + ".png") for k in now],
                        "lidar": [os.path.join(route_dir, "lidar", k + ".npy") for k in now],
                        "radar": [os.path.join(route_dir, "radar", k + ".npy") for k in now],
                        "vectormap": [os.path.join(route_dir, "vectormap", k + ".npy") for k in now],
                        "x": [m["x"] for m in meas], "y": [m["y"] for m in meas], "theta": thetas,
                        "x_command": cur["x_command"], "y_command": cur["y_command"], "steer": cur["steer"],
                        "throttle": cur["throttle"], "brake": cur["brake"], "command": cur["command"], "velocity": cur["speed"],
                    })

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        from PIL import Image
        f = self.frames[i]
        S = self.seq_len
        ego = S - 1
        lanes = []
        for t in range(S):
            vm, j = f["vectormap"][t], i
            while not os.path.exists(vm):  # frames recorded without a lane file borrow a neighbour's (dataloader.py:199-207)
                j = j - 1 if j - 1 >= 0 else j + 1
                vm = self.frames[j]["vectormap"][t]
            lanes.append(torch.from_numpy(np.load(vm)))
        rgbs = [torch.from_numpy(np.array(Image.open(p).convert("RGB"), dtype=np.uint8)) for p in f["rgb"]]
        sample = {
            "vectormaps": lanes,
            "radar": [radar_to_size(np.load(p)) for p in f["radar"]],
            "waypoints": local_waypoints(f["x"], f["y"], f["theta"], ego),
            "target_point": local_target_point(f["x_command"], f["y_command"], f["x"][ego], f["y"][ego], f["theta"][ego]),
            "steer": f["steer"], "throttle": f["throttle"], "brake": f["brake"], "command": f["command"], "velocity": f["velocity"],
        }
        if S == 1:
            sample["rgb_u8"] = rgbs[0]
            sample["lidar_pts"] = torch.from_numpy(np.load(f["lidar"][0]).astype(np.float32)[:, :4])
        else:
            sample["rgb_u8"] = torch.stack(rgbs, 0)
            sweeps = []
            for t in range(S):
                raw = np.load(f["lidar"][t])
                xyz = raw[..., :3].astype(np.float64)
                xyz[:, 1] *= -1                                              # dataloader.py:227
                xyz = ego_transform(xyz, np.pi / 2 - f["theta"][t], -f["x"][t], -f["y"][t],
                                    np.pi / 2 - f["theta"][ego], -f["x"][ego], -f["y"][ego])
                pts = np.zeros((len(xyz), 4), np.float64)
                pts[:, :3] = xyz
                if raw.shape[1] > 3:
                    pts[:, 3] = raw[:, 3]
                sweeps.append(pts)
            sample["lidar_pts"] = sweeps          # float64 [N_t, 4] each, already in the ego frame
            sample["lidar_in_ego_frame"] = True
            sample["lidar_reference_frame"] = ego
        if all(os.path.exists(p) for p in f["map"]):
            sample["maps"] = [torch.from_numpy(np.ascontiguousarray(np.transpose(np.asarray(Image.open(p)), (2, 0, 1)))) for p in f["map"]]
        sample["radar_adj"] = radar_adjacency(sample["radar"][0])
        return sample


def _read_json(path):
    import json
    with open(path) as f:
        return json.load(f)


FAR_POINT = 1.0e6  # padding x coordinate: outside every histogram bin (np.histogramdd ignores it the same way)


def collate_raw(samples):
    """RawFrameStore samples -> batch with `rgb_u8` [B,H,W,3] u8 and `lidar_pts` [B,Nmax,4] padded with far points;
    everything else as `collate`."""
    skip = ("rgb_u8", "lidar_pts", "lidar_in_ego_frame", "lidar_reference_frame")
    rest = [{k: v for k, v in s.items() if k not in skip} for s in samples]
    out = collate(rest)
    out["rgb_u8"] = torch.stack([s["rgb_u8"] for s in samples], 0)          # [B, H, W, 3] or, seq_len > 1, [B, S, H, W, 3]
    several = isinstance(samples[0]["lidar_pts"], (list, tuple))
    sweeps = [s["lidar_pts"] if several else [s["lidar_pts"]] for s in samples]
    S = len(sweeps[0])
    nmax = max(int(p.shape[0]) for sw in sweeps for p in sw)
    nmax = (nmax + 1023) // 1024 * 1024  # few distinct shapes -> few buffer sets / graph captures downstream
    pts = torch.zeros(len(samples), S, nmax, 4, dtype=torch.float32)
    pts[:, :, :, 0] = FAR_POINT
    for b, sw in enumerate(sweeps):
        for t, p in enumerate(sw):
            p = torch.as_tensor(np.asarray(p)).to(torch.float32)
            pts[b, t, :p.shape[0], :p.shape[1]] = p
    out["lidar_pts"] = pts if several else pts[:, 0]                      # [B, S, N, 4] (ego frame already) / [B, N, 4] (raw)
    out["lidar_in_ego_frame"] = bool(samples[0].get("lidar_in_ego_frame", False))
    return out


def stage_raw_batch(data, device, config, variant="vec", non_blocking=True):
    """collate_raw batch -> (Engine input dict, gt_waypoints) on `device`: the frames stay uint8 / XYZI; the y flip of
    dataloader.py:233 is requested from the splat kernel."""
    dev = torch.device(device)
    to = lambda t, dt=None: torch.as_tensor(t).to(dt or torch.as_tensor(t).dtype).to(dev, non_blocking=non_blocking)
    if variant != "img" and config.seq_len > 1:
        # with several frames per sample the ego frame is the LAST one (waypoints, target point, sweeps moved into it); lanes and
        # radar below are frame 0's.  The vector-map / radar models cannot run seq_len > 1 anyway (Engine.__init__, as the
        # reference: model_vec.py:226) - refuse here too instead of pairing frame-0 lanes with last-frame labels
        raise NotImplementedError("raw batches with seq_len > 1 exist for the image-map model only (variant 'img')")
    lane, lane_num, _ = data["vectormaps"][0]
    rgb, pts = to(data["rgb_u8"]), to(data["lidar_pts"])
    if rgb.dim() == 5:   # seq_len > 1: a sample's frames become consecutive batch entries (model_vec.py:506-508)
        rgb, pts = rgb.flatten(0, 1), pts.flatten(0, 1)
    inp = {
        "rgb_u8": rgb, "lidar_pts": pts, "lidar_flip_y": not data.get("lidar_in_ego_frame", False),
        "target_point": to(torch.stack(list(data["target_point"]), dim=1), torch.float32),
        "velocity": to(data["velocity"], torch.float32),
    }
    if variant == "img":
        maps = [to(m).to(torch.float32) for m in data["maps"][:config.seq_len]]
        inp["map"] = maps[0] if len(maps) == 1 else torch.stack(maps, dim=1).flatten(0, 1)
    else:
        inp["lane"] = to(lane, torch.float32)
        inp["lane_num"] = to(lane_num, torch.int32)
    if variant == "rad":
        inp["radar"] = to(data["radar"][0], torch.float32)
        inp["radar_adj"] = to(data["radar_adj"], torch.float32)
    wps = data["waypoints"]
    n = config.seq_len
    gt = torch.stack([torch.stack(list(wps[i]), dim=1) for i in range(n, len(wps))], dim=1)
    return inp, to(gt, torch.float32)


def preprocess_routes(store, out_dir, device, batch_size=16):
    """Phase 1 on the GPU (run_steps/phase1_preprocess_data.py:42-48): RawFrameStore -> one PRE_Data pickle per frame, with
    the camera crop and the LiDAR histogram produced by the ingest kernels.  Returns the number of files written."""
    from . import ops
    os.makedirs(out_dir, exist_ok=True)
    dev = torch.device(device)
    n = 0
    for start in range(0, len(store), batch_size):
        samples = [store[i] for i in range(start, min(len(store), start + batch_size))]
        batch = collate_raw(samples)
        pts = batch["lidar_pts"].to(dev)
        S = 1
        if pts.dim() == 4:   # seq_len > 1: [B, S, N, 4], already in the ego frame (no y flip on the device)
            S = pts.shape[1]
            pts = pts.flatten(0, 1)
        bev = ops.lidar_splat(pts, torch.empty(len(samples) * S, 256, 256, 2, device=dev), flip_y=not batch["lidar_in_ego_frame"])
        bev = bev.permute(0, 3, 1, 2).cpu().numpy().reshape(len(samples), S, 2, 256, 256)
        for k, s in enumerate(samples):
            frames = s["rgb_u8"].numpy()
            frames = frames[None] if frames.ndim == 3 else frames
            H, W = frames.shape[1:3]
            rec = {key: s[key] for key in ("vectormaps", "radar", "waypoints", "target_point", "steer", "throttle", "brake", "command",
                                           "velocity") if key in s}
            rec["fronts"] = [torch.from_numpy(np.ascontiguousarray(np.transpose(fr[H // 2 - 128:H // 2 + 128, W // 2 - 128:W // 2 + 128], (2, 0, 1))))
                             for fr in frames]
            rec["lidars"] = [np.ascontiguousarray(bev[k, t]) for t in range(S)]
            rec["maps"] = s.get("maps", [torch.zeros(3, 256, 256, dtype=torch.uint8) for _ in range(S)])
            with open(os.path.join(out_dir, "%d.pkl" % (start + k)), "wb") as fd:
                pickle.dump(rec, fd)
            n += 1
    return n
