#!/usr/bin/env python
"""MMFN training-step benchmark on MI355X (BASELINE.json metric: train samples/sec).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL gradient all-reduce)

Workload (BASELINE.json configs[1]): full MMFN "vec" variant, fp32, batch
32 per GPU, synthetic sensor data resident in HBM (u8 300x400x3 camera frames, 16384-point
XYZI LiDAR sweeps, 64x10x5 lane polylines), random-init weights.  One step = on-GPU ingest
(crop/normalise, BEV splat) + forward + L1 loss + backward + AdamW — nothing is skipped.

Prints ONE JSON line with the contract fields plus
  roofline      the fp32-MFMA GEMM/implicit-conv kernel family (99 % of the FLOPs): algorithmic
                FLOPs of its launches / their summed duration, timed with HIP events on the
                launch stream in extra instrumented steps after the timed region,
  cpu_baseline  the CPU oracle (oracle/, a plain-PyTorch restatement pinned to the reference)
                timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak (AMD's 5 PF headline includes 2:1 sparsity)
ALGO_GFLOP_PER_SAMPLE = {"vec": 106.6, "img": 112.4, "rad": 117.7, "image-only": 28.4}


# BASELINE.json configs -> flags (`--config NAME`); vec32 is what `python bench.py` runs without flags
PRESETS = {
    "vec32": ("configs[1]: full MMFN vec, fp32, batch 32, 1 GPU", dict()),
    "bf16": ("configs[2] per-GPU arithmetic: bf16 training mode, batch 32/GPU (add --gpus 8 for the 8 x 32 = 256 run)", dict(dtype="bf16")),
    "img128": ("configs[3]: ResNet-34 camera branch alone, batch 128", dict(workload="image-only", batch=128)),
    "rad16": ("configs[4]: rad variant (4 modalities), 65536-point LiDAR, batch 16", dict(variant="rad", batch=16, n_lidar=65536)),
}


def synth_inputs(B, device, seed, lanes=64, n_lidar=16384, variant="vec", lane_format="10x5", seq_len=1, n_views=1):
    """B samples; with seq_len / n_views > 1 a sample's n_views*seq_len camera frames, seq_len sweeps and seq_len maps are
    consecutive batch entries (torch.stack(list, dim=1).view(bz * n, ...), model_vec.py:506-508)."""
    g = torch.Generator().manual_seed(seed)
    Bi, Bs = B * n_views * seq_len, B * seq_len
    rgb = torch.randint(0, 256, (Bi, 300, 400, 3), generator=g, dtype=torch.uint8)
    pts = torch.empty(Bs, n_lidar, 4)
    pts[..., 0:2] = torch.rand(Bs, n_lidar, 2, generator=g) * 40.0 - 20.0
    pts[..., 2] = torch.rand(Bs, n_lidar, generator=g) * 4.0 - 3.0
    pts[..., 3] = torch.rand(Bs, n_lidar, generator=g)
    pts[:, n_lidar - n_lidar // 16:, 0] = 1e6
    lane = torch.zeros(B, lanes, 10, 5)
    lane[..., 0:2] = torch.randn(B, lanes, 10, 2, generator=g) * 8.0
    lane[..., 2:5] = torch.randint(0, 2, (B, lanes, 10, 3), generator=g).float()
    lane_num = torch.randint(1, lanes + 1, (B,), generator=g)
    lane_num[0] = lanes
    if lane_format == "19x8":  # north_star's perf-only variant: polylines already vectorised, 19 vectors x 8 features
        lane = torch.zeros(B, lanes, 19, 8)
        lane[..., 0:4] = torch.randn(B, lanes, 19, 4, generator=g) * 8.0
        lane[..., 4:8] = torch.randint(0, 2, (B, lanes, 19, 4), generator=g).float()
    for i in range(B):
        lane[i, int(lane_num[i]):] = 0.0
    inp = {
        "rgb_u8": rgb, "lidar_pts": pts, "lane": lane, "lane_num": lane_num.to(torch.int32),
        "target_point": torch.randn(B, 2, generator=g) * 10.0, "velocity": torch.rand(B, generator=g) * 8.0,
    }
    gt = torch.randn(B, 4, 2, generator=g) * 5.0
    if variant == "img":  # raster map instead of lanes (model_img.py:337: not normalised)
        del inp["lane"], inp["lane_num"]
        inp["map"] = torch.randint(0, 256, (Bs, 3, 256, 256), generator=g, dtype=torch.uint8).float()
    if variant == "rad":
        radar = torch.randn(B, 81, 5, generator=g)
        radar[..., 3] = radar[..., 3].abs() + 0.5
        inp["radar"] = radar
        inp["radar_adj"] = radar[:, None, :, 1] - radar[:, :, None, 1]  # adj[i, j] = r[j, 1] - r[i, 1] (dataloader.py:381-384)
    return {k: v.to(device).contiguous() for k, v in inp.items()}, gt.to(device)


class ImageBranchOnly(object):
    """BASELINE.json configs[3]: the ResNet-34 camera branch alone (ingest -> stem -> layer1..4 -> global average
    pool), forward + backward (dgrad + wgrad + BN), no fusion transformers and no optimizer: a conv/MFMA run."""

    def __init__(self, eng, rgb_u8):
        from mmfn_amd import ops
        self.eng, self.ops, self.rgb = eng, ops, rgb_u8
        self.B = rgb_u8.shape[0]
        self.gseed = torch.full((self.B, 512), 1.0 / (self.B * 512), device=rgb_u8.device)

    def __call__(self):
        eng, ops, B = self.eng, self.ops, self.B
        ctx = eng._ctx(B, True)
        bufs = ctx.bufs
        x = ops.ingest_rgb_u8(self.rgb, bufs.get("in.img", (B, 256, 256, 3)))
        f = eng.img.stem_fwd(ctx, x)
        for li in range(1, 5):
            f = eng.img.layer_fwd(ctx, li, f)
        pooled = ops.gap_sum_fwd([f], bufs.get("fused", (B, 512)))
        g = bufs.get("G3.0", f.shape, f.dtype)   # (bf16 in the bf16 mode)
        ops.gap_sum_bwd(self.gseed, [g])
        for li in range(4, 0, -1):
            g = eng.img.layer_bwd(ctx, li, g)
        eng.img.stem_bwd(ctx, g)
        return pooled



def attention_roofline(eng, B, dev, iters=20):
    """The fusion-attention kernels of the widest transformer (head size n_embd / n_head = 128, T tokens) timed alone on the stream the
    step launches them on, ONE normalisation: the FLOPs their MFMAs execute (forward 2 products, backward 3 + 4 incl. the recomputed
    S and dP) / kernel time (HIP events around `iters` back-to-back launches) / the nominal fp32 MFMA peak."""
    import math
    from mmfn_amd import ops
    gpt = eng.gpts[-1]
    T, C, NH = gpt.T, gpt.C, gpt.nh
    HS = C // NH
    qkv = torch.randn(B * T, 3 * C, device=dev)
    dO = torch.randn(B * T, C, device=dev)
    o, dqkv = torch.empty(B * T, C, device=dev), torch.empty(B * T, 3 * C, device=dev)
    lse, delta = torch.empty(B, NH, T, device=dev), torch.empty(B, NH, T, device=dev)
    rng = torch.tensor([5, 1], dtype=torch.int64, device=dev)
    p, sc = float(eng.cfg.attn_pdrop), 1.0 / math.sqrt(HS)
    fwd = lambda: ops.attention_fwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, C, lse, B, T, NH, HS, sc, drop_p=p, rng_state=rng, rng_stream=3)
    bwd = lambda: ops.attention_bwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, dO, C, lse, delta, dqkv[:, C:], dqkv, dqkv[:, 2 * C:], 3 * C,
                                    B, T, NH, HS, sc, drop_p=p, rng_state=rng, rng_stream=3)
    out = {"head_size": HS, "tokens": T, "heads": NH, "batch": B, "dropout": p, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
           "normalisation": "executed MFMA FLOPs / kernel time (HIP events over %d back-to-back launches) / nominal fp32 MFMA peak" % iters}
    unit = 2.0 * T * T * C * B   # one T x T x head-size product over all heads and samples
    for name, fn, products in (("fwd", fwd, 2), ("bwd", bwd, 7)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        tf = products * unit / (us * 1e-6) / 1e12
        out[name] = {"us": round(us, 1), "achieved": round(tf, 1), "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 3),
                     "kernels": ("attn_wg_fwd" if name == "fwd" else "attn_wg_dq + attn_wg_dkv") if T in (64, 128, 192, 256)
                     else ("attn_fwd (attention.hip tile kernels)" if name == "fwd" else "attn_bwd_dq + attn_bwd_dkv (attention.hip tile kernels)")}
    return out


def also_config(args, name):
    """Another BASELINE configuration measured by the SAME driver run: a second bench process started after the fp32 line's
    timed region (this process keeps its buffers; 288 GB holds both), same steps / warm-up.  `name` is a PRESETS key: "bf16"
    (configs[2] per-GPU arithmetic), "img128" (configs[3]), "rad16" (configs[4]).  Returns the fields of its JSON line that
    matter, or {"error": ...} - the fp32 line never depends on it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--no-cpu-baseline", "--no-also"]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
        line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            return {"error": "bench.py --config %s exited %d: %s" % (name, out.returncode, out.stderr.decode()[-400:])}
        rec = json.loads(line[-1])
    except Exception as exc:   # noqa: BLE001 - whatever happens there, the fp32 line is printed
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "loss", "roofline", "loss_vs_oracle")
    rec = {k: rec[k] for k in keep if k in rec}
    rec["config"] = PRESETS[name][0]
    rec["command"] = "python bench.py --config %s --steps %d --warmup %d" % (name, args.steps, args.warmup)
    return rec


def csrc_digest():
    """sha256 over the kernel sources (mmfn_amd/csrc/*, sorted by name): what a committed PMC profile is valid for.  The GPU box
    has no .git, so staleness of profiles/*_traffic.json is judged by this digest (tools/summarize_profile.py stamps it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mmfn_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def default_like_bf16(args, B):
    """True when the run is the workload the committed bf16 traffic profile was taken on."""
    return (args.workload == "train" and args.variant == "vec" and B == 32 and args.n_lidar == 16384 and args.lane_format == "10x5")


def usable_cores():
    """Cores this process may actually use (affinity mask capped by the cgroup CPU quota): mmfn_amd.data.usable_cores."""
    from mmfn_amd.data import usable_cores as _uc
    return _uc()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_train_rate(batch, warmup, steps, threads):
    """Median step time of the oracle's train step (zero-grad + fwd + L1 + bwd + AdamW, fp32, dropout 0.1 active)."""
    from oracle import fixtures, harness
    model = harness.build_oracle("vec", dropout=0.1)
    b = fixtures.synthetic_batch(batch, "vec", seed=42, lanes=64)
    args = harness.forward_args(b, "vec")
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    model.train()
    times = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        for p in model.parameters():
            p.grad = None
        loss = harness.l1_waypoint_loss(model(*args), b["gt_wp"])
        loss.backward()
        opt.step()
        if i >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    return batch / times[len(times) // 2], times[len(times) // 2]


def cpu_baseline(big_batch=32, small_batch=2, warmup=2, steps=5):
    """the reference's CPU PyTorch path (here: the oracle, its validated restatement) timed on this
    box's host cores - same step as the GPU (fwd + L1 + bwd + AdamW, fp32), all usable cores, B=2 (BASELINE configs[0]) and
    the per-GPU batch (32), median of `steps` after `warmup` warm-ups, anomaly mode off.  `value` is the batch-32 figure
    (the like-for-like comparison with the GPU line); the batch-2 figure rides along."""
    threads = usable_cores()
    torch.set_num_threads(threads)
    v2, t2 = _cpu_train_rate(small_batch, warmup, steps, threads)
    v32, t32 = _cpu_train_rate(big_batch, warmup, steps, threads)
    return {"value": round(v32, 3), "unit": "samples/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
            "value_batch2": round(v2, 3),
            "sample": "oracle train step (vec, fp32, %d torch threads), median of %d steps after %d warm-ups: batch %d %.2f s/step, "
                      "batch %d %.2f s/step" % (threads, steps, warmup, big_batch, t32, small_batch, t2)}


def oracle_batch_from_inputs(inp, variant):
    """Device-resident engine inputs (synth_inputs) -> the CPU batch dict oracle.harness.forward_args takes."""
    B = inp["target_point"].shape[0]
    cpu = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    return {"rgb_u8": cpu["rgb_u8"], "lidar_pts": cpu["lidar_pts"], "target_point": cpu["target_point"], "velocity": cpu["velocity"],
            "lane": cpu.get("lane", torch.zeros(B, 1, 10, 5)), "lane_num": cpu.get("lane_num", torch.ones(B)).long(),
            "map_u8": cpu.get("map", torch.zeros(B, 3, 256, 256)), "radar": cpu.get("radar", torch.zeros(B, 81, 5)),
            "radar_adj": cpu.get("radar_adj", torch.zeros(B, 81, 81))}


def loss_vs_oracle(net, eng, inp, gt, variant):
    """Waypoint L1 loss of the HIP path vs the CPU oracle on THE BENCHED BATCH with the current weights (train-mode BatchNorm,
    dropout switched off on both sides because the reference's RNG stream cannot be reproduced).  Outside the timed region;
    BatchNorm running statistics are restored afterwards."""
    from oracle import harness
    L = net._layout
    cfg = eng.cfg
    saved_p = (cfg.embd_pdrop, cfg.attn_pdrop, cfg.resid_pdrop)
    saved_buf = (L.buffers_flat.clone(), L.counters_flat.clone())
    rad_p = None if eng.rad is None else eng.rad.p
    cfg.embd_pdrop = cfg.attn_pdrop = cfg.resid_pdrop = 0.0
    if eng.rad is not None:
        eng.rad.p = 0.0
    try:
        _, loss = eng.forward(inp, True, gt)
        hip = float(loss.item())
    finally:
        cfg.embd_pdrop, cfg.attn_pdrop, cfg.resid_pdrop = saved_p
        if eng.rad is not None:
            eng.rad.p = rad_p
        L.buffers_flat.copy_(saved_buf[0])
        L.counters_flat.copy_(saved_buf[1])
    oracle = harness.build_oracle(variant, dropout=0.0, lane_channels=getattr(eng.cfg, "lane_channels", 7))
    oracle.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
    batch = oracle_batch_from_inputs(inp, variant)
    torch.set_num_threads(usable_cores())
    oracle.train()
    with torch.no_grad():
        ref = float(harness.l1_waypoint_loss(oracle(*harness.forward_args(batch, variant)), gt.cpu()).item())
    return {"hip": round(hip, 7), "oracle": round(ref, 7), "abs_diff": float("%.3g" % abs(hip - ref)), "tolerance": 1e-4,
            "note": "benched batch, current weights, train-mode BatchNorm, dropout off on both sides"}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` without an outer launcher: start the N ranks ourselves, one process per GPU, through
    torch.distributed.run on this node (what the driver's own multi-GPU command line does), and pass the ranks' output and
    exit code through.  Rank 0 prints the one JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, usable_cores() // n))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def numa_cpus_of_gpu(index):
    """CPUs of the NUMA node the GPU's PCIe function hangs off, or None when the platform does not say."""
    try:
        props = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        return cpus
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def pin_rank(local_rank, world, gpu_index):
    """One disjoint slice of the usable cores per rank, taken from the GPU's own NUMA node when that is known: the replay
    thread, RCCL's proxy thread and the launcher's OpenMP workers of eight ranks otherwise migrate over the whole host and share
    cores.  Returns a description for the JSON line.  MMFN_BENCH_PIN=0 switches it off."""
    if os.environ.get("MMFN_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    near = numa_cpus_of_gpu(gpu_index)
    pool = [c for c in allowed if near is None or c in near]
    same_node_ranks = world
    if near is not None and len(pool) >= 2:
        # ranks whose GPUs share this NUMA node split ITS cores; without topology every rank splits the whole mask
        peers = [r for r in range(world) if numa_cpus_of_gpu(r if not os.environ.get("MMFN_BENCH_SINGLE_DEVICE") else 0) == near]
        same_node_ranks, slot = max(1, len(peers)), (peers.index(local_rank) if local_rank in peers else local_rank % max(1, len(peers)))
    else:
        pool, slot = allowed, local_rank
    per = max(1, len(pool) // same_node_ranks)
    mine = pool[slot * per:(slot + 1) * per] or pool
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return {"cpus": len(mine), "first": mine[0], "numa_aware": near is not None}


def all_ranks_agree(dist, dev, ok):
    from mmfn_amd.comm import all_ranks_agree as _f
    return _f(dist, dev, ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 100 x ~32 ms: a timed region of seconds, not of a few scheduler quanta
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--variant", default="vec", choices=["vec", "img", "rad"])
    ap.add_argument("--workload", default="train", choices=["train", "image-only"],
                    help="train = full step (BASELINE configs[1]); image-only = ResNet-34 branch fwd+bwd (configs[3])")
    ap.add_argument("--n-lidar", type=int, default=16384, help="LiDAR points per sample (configs[4]: 65536)")
    ap.add_argument("--seq-len", type=int, default=1, help="frames per sample and modality (image-map model only; the reference trains with 1)")
    ap.add_argument("--n-views", type=int, default=1, help="camera views per frame (the reference trains with 1)")
    ap.add_argument("--lane-format", default="10x5", choices=["10x5", "19x8"],
                    help="10x5 = the reference's lane nodes (parity format, default); 19x8 = north_star's perf-only pre-vectorised "
                         "polylines [B,64,19,8] (VectorNet with lane_channels=8; no reference checkpoint has this shape)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "bf16-operands"],
                    help="f32 = the parity path (headline).  bf16 = the bf16 training mode (BASELINE configs[2]): bf16 activations, "
                         "saved tensors and weight shadows in HBM, bf16 MFMA with fp32 accumulation, fp32 statistics / master weights "
                         "/ gradients / optimizer.  bf16-operands = round 2's mode: fp32 tensors in HBM, GEMM operands rounded to "
                         "bf16 on their way into LDS")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"],
                    help="data-parallel runs: element type of the gradient buckets on the wire.  f32 (default) is what the reference's DDP "
                         "exchanges, also under autocast; bf16 is an opt-in (210 instead of 419 MB per step, rounding at every hop)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-check", action="store_true", help="skip the loss_vs_oracle block (one CPU oracle forward)")
    ap.add_argument("--single-stream", action="store_true", help="disable encoder-branch concurrency (profiling runs)")
    ap.add_argument("--profile-steps", type=int, default=2, help="instrumented steps for the roofline block")
    ap.add_argument("--breakdown", action="store_true", help="print the per-operation table of the roofline block's instrumented steps to stderr")
    ap.add_argument("--no-attention-roofline", action="store_true", help="skip roofline.attention (the widest transformer's attention kernels timed alone)")
    ap.add_argument("--no-also", action="store_true",
                    help="default one-GPU run only: do not append the other BASELINE configurations' records (`also.bf16`, `also.img128`, "
                         "`also.rad16`: further bench processes after the fp32 line's timed region)")
    ap.add_argument("--config", default=None, choices=sorted(PRESETS),
                    help="BASELINE.json configuration presets (one flag per config): " + "; ".join("%s = %s" % (k, v[0]) for k, v in sorted(PRESETS.items())))
    args = ap.parse_args()
    if args.config:   # a preset only fills in the flags the command line left at their defaults
        for k, v in PRESETS[args.config][1].items():
            if getattr(args, k) == ap.get_default(k):
                setattr(args, k, v)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no outer launcher: start the ranks ourselves (the same command line works under torch.distributed.run too)
        single = bool(os.environ.get("MMFN_BENCH_SINGLE_DEVICE"))
        if not single and torch.cuda.device_count() < args.gpus:
            raise SystemExit("--gpus %d: only %d GPU(s) visible on this node" % (args.gpus, torch.cuda.device_count()))
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # host-side tensor ops (synthetic inputs, parameter init) must not fan out over every logical CPU of the box in
    # every rank: under a cgroup quota the spinning OpenMP workers get the whole process throttled
    torch.set_num_threads(max(1, min(8, usable_cores() // max(1, world))))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    gpu_index = local_rank
    if os.environ.get("MMFN_BENCH_SINGLE_DEVICE"):  # CI on a 1-GPU box: all ranks share cuda:0 (with gloo, see below)
        gpu_index = 0
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    pinned = pin_rank(local_rank, world, gpu_index) if world > 1 else None
    local_rank = gpu_index
    dist = None
    if world > 1:
        # a multi-rank run that stops making progress (a collective some rank never joins) says where it stood instead of sitting
        # there until the launcher is killed from outside: all threads' stacks to stderr, then exit 124
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get("MMFN_BENCH_WATCHDOG_S", "900")), exit=True)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MMFN_DIST_BACKEND", "gloo" if os.environ.get("MMFN_BENCH_SINGLE_DEVICE") else "nccl")
        if backend == "nccl":   # nccl == RCCL over xGMI on ROCm
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mmfn_amd import ops
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN, MMFNImg, MMFNRad
    from mmfn_amd.parallel import DataParallel

    torch.manual_seed(42)  # init_torch(): run_steps/utils.py:77-84
    net = {"vec": MMFN, "img": MMFNImg, "rad": MMFNRad}[args.variant](GlobalConfig(gemm_dtype="bf16" if args.dtype == "bf16-operands" else "f32", act_dtype="bf16" if args.dtype == "bf16" else "f32",
                                                                       lane_channels=8 if args.lane_format == "19x8" else 7, seq_len=args.seq_len, n_views=args.n_views), dev)
    net.train()
    B = args.batch
    inp, gt = synth_inputs(B, dev, seed=42 + rank, n_lidar=args.n_lidar, variant=args.variant, lane_format=args.lane_format,
                           seq_len=args.seq_len, n_views=args.n_views)
    several_frames = (args.seq_len, args.n_views) != (1, 1)
    if several_frames:   # the oracle check and the CPU baseline below are wired for the one-frame workloads
        args.no_oracle_check = args.no_cpu_baseline = True
    comm_capi, transport_note, dp = None, None, None
    if world > 1:
        # gradient buckets through the C ABI (libmmfn_comm.so -> RCCL on our own stream: capturable, so the whole data-parallel
        # step is ONE hipGraph) whenever the library loads and its communicator passes a self-test on every rank; otherwise
        # torch's ProcessGroup with the step cut at the bucket boundaries (mmfn_amd.parallel.connect; MMFN_DP_TRANSPORT)
        from mmfn_amd.parallel import connect
        dp, transport_note = connect(net, dist, grad_dtype=args.grad_dtype)
        comm_capi = dp.comm
        dp.broadcast_parameters()
    eng = net._engine_for()
    if args.single_stream:
        eng.multi_stream = False

    image_only = args.workload == "image-only"
    if image_only:
        if world > 1:
            raise SystemExit("the image-only ablation is a single-GPU run")
        step = ImageBranchOnly(eng, inp["rgb_u8"])
    else:
        def step():
            return eng.train_step(inp, gt, lr=1e-4, dp=dp)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def lock_step():
        """True when the trained parameters are bit-identical on every rank - i.e. every gradient all-reduce so far delivered
        the same sum everywhere (all ranks start from rank 0's broadcast and apply the same AdamW)."""
        if dist is None:
            return True
        L = net._layout
        chk = torch.sum(L.params[:L.tail].view(torch.int32), dtype=torch.int64).reshape(1)
        if dist.get_backend() != "nccl":
            chk = chk.cpu()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool((lo == hi).item())

    def to_torch_transport(why):
        """Give up on the C-ABI transport (wrong sums, failed capture): fresh DataParallel over torch.distributed, parameters
        re-broadcast from rank 0."""
        nonlocal dp, comm_capi, transport_note
        if rank == 0:
            sys.stderr.write("C-ABI RCCL transport dropped (%s); continuing on torch.distributed\n" % why)
        comm_capi, transport_note = None, "fallback: " + why
        dp = DataParallel(net, dist, grad_dtype=args.grad_dtype)
        dp.broadcast_parameters()

    # two eager steps size every buffer, then (optionally) capture one step into a hipGraph
    step(); step()
    torch.cuda.synchronize()
    if comm_capi is not None and not lock_step():
        to_torch_transport("ranks diverged after two eager steps")
        step(); step()
        torch.cuda.synchronize()
    runner = step
    graph = None
    if not args.no_graph and image_only:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            loss_buf = step()
        runner = graph.replay
    elif not args.no_graph:
        # single GPU: one hipGraph.  Data parallel: one hipGraph with the RCCL all-reduces captured inside (C-ABI transport), or
        # four graphs cut at the backward-stage boundaries with the bucket all-reduces in between (torch.distributed)
        from mmfn_amd.parallel import GraphedStep

        def try_capture():
            try:
                g, err = GraphedStep(eng, dp, inp, gt, lr=1e-4, warm=0), None
            except Exception as exc:
                g, err = None, "%s: %s" % (type(exc).__name__, exc)
                torch.cuda.synchronize()
            if dist is not None and not all_ranks_agree(dist, dev, g is not None):
                g, err = None, err or "capture failed on another rank"
            return g, err

        graph, err = try_capture()
        if graph is not None and comm_capi is not None:
            for _ in range(2):
                graph()
            torch.cuda.synchronize()
            if not lock_step():
                graph, err = None, "ranks diverged after replaying the single-graph step"
        if graph is None and comm_capi is not None:
            to_torch_transport(err)
            step()
            torch.cuda.synchronize()
            graph, err = try_capture()
        if graph is not None:
            runner = graph
        else:  # keep the run alive: eager launches give the same numbers' meaning, only slower
            sys.stderr.write("rank %d: hipGraph capture failed (%s); falling back to eager launches\n" % (rank, err))
            torch.cuda.synchronize()
            step()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    comm = None
    if dp is not None:
        in_step = lock_step()
        single_graph = graph is not None and getattr(graph, "single_graph", False)
        if single_graph:
            # the collectives live inside the graph, so the exposed communication is measured as the difference to the same
            # step without them: a second capture with dp=None (outside the timed region; the ranks drift apart from here on,
            # nothing after this needs them in lock step)
            n_probe = max(3, min(10, args.steps))
            ref = GraphedStep(eng, None, inp, gt, lr=1e-4, warm=0)
            for _ in range(2):
                ref()
            barrier()
            t1 = time.perf_counter()
            for _ in range(n_probe):
                ref()
            torch.cuda.synchronize()
            alone = (time.perf_counter() - t1) / n_probe * 1e3
            ex_val, ex_how = max(0.0, ms_per_step - alone), "ms_per_step minus the same captured step without collectives (%.3f ms)" % alone
        else:
            # time the compute stream waits for the gradient all-reduces at the end of the backward (what the overlap does not
            # hide), over a few extra steps outside the timed region
            dp.measure_exposed = True
            for _ in range(max(1, min(3, args.steps))):
                runner()
            ex_val, ex_how = dp.exposed_ms() or 0.0, "HIP events around the wait for the reductions before AdamW"
            dp.measure_exposed = False
        ex = torch.tensor([ex_val], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)   # max over ranks
        nbytes = dp.bytes_per_step()
        comm = {"backend": dist.get_backend(), "library": "RCCL over xGMI" if dist.get_backend() == "nccl" else dist.get_backend(),
                "transport": ("mmfn_allreduce_sum_%s (C ABI, libmmfn_comm.so)" % dp.grad_dtype) if comm_capi is not None else "torch.distributed",
                "gradient_dtype_on_the_wire": dp.grad_dtype,
                "ranks": dist.get_world_size(), "allreduce_bytes_per_step": nbytes, "buckets": dp.n_buckets(),
                "graphs_per_step": (1 if single_graph else graph.recorder.n_graphs) if graph is not None else 0,
                "exposed_ms_per_step": round(float(ex.item()), 3), "exposed_method": ex_how,
                "ranks_in_lock_step": in_step, "cpu_pinning": pinned}
        if transport_note:
            comm["note"] = transport_note
    loss_val = None if image_only else float(eng._bufs_for(B).get("head.loss", (1,)).item())
    workload = ("full MMFN %s (ResNet34 img + ResNet18 LiDAR-BEV + %s -> 4 GPT fusion -> GRU), train step fwd+L1+bwd+AdamW, "
                "batch %d/GPU, 400x300x3 u8 RGB + %d-pt LiDAR + %s"
                % (args.variant, {"vec": "VectorNet", "img": "ResNet34 raster map", "rad": "VectorNet + radar GAT"}[args.variant], B,
                   args.n_lidar, "256x256x3 raster map" if args.variant == "img" else ("64x19x8 pre-vectorised polylines" if args.lane_format == "19x8" else "64x10x5 lanes")
                   + (" + 81x5 radar" if args.variant == "rad" else "")))
    if several_frames:
        workload += "; seq_len %d, n_views %d: %d camera + %d LiDAR + %d map frames per sample" % (
            args.seq_len, args.n_views, args.seq_len * args.n_views, args.seq_len, args.seq_len)
    if image_only:
        workload = "ResNet-34 camera branch alone, fwd+bwd (no optimizer), batch %d, 400x300x3 u8 RGB" % B

    result = {
        "metric": "image-branch fwd+bwd samples/sec" if image_only else "train samples/sec (RGB+LiDAR+vec-map fusion)", "value": round(value, 2), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype != "f32" else "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "global_batch": B * world, "parallelism": "dp%d" % world, "hipgraph": graph is not None,
                   "branch_streams": 1 if args.single_stream else eng.n_lanes},
        "loss": None if loss_val is None else round(loss_val, 6),
    }
    if comm is not None:
        result["comm"] = comm
    if rank == 0:
        # ---- roofline of the dominant kernel family (fp32 MFMA GEMM / implicit conv)
        prof = ops.GemmProfiler()
        ops.set_gemm_profiler(prof)
        eng.multi_stream = False  # time each launch alone: with branch concurrency on, kernels of other
        #                           streams share the CUs and per-launch durations are not comparable
        for _ in range(max(1, args.profile_steps)):
            step() if image_only else eng.train_step(inp, gt, lr=1e-4, dp=None)
        torch.cuda.synchronize()
        ops.set_gemm_profiler(None)
        eng.multi_stream = not args.single_stream
        n_launch, flops, ms = prof.summary()
        executed = prof.executed_flops()
        traffic, traffic_src, traffic_commit, traffic_digest = None, None, None, None
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
        default_workload = (not image_only and args.variant == "vec" and B == 32 and args.n_lidar == 16384 and args.dtype == "f32"
                            and args.lane_format == "10x5")
        if tfiles and default_workload:  # PMC-derived HBM bytes per launch of this kernel family (tools/profile_round.sh)
            rec = json.load(open(tfiles[-1]))
            traffic, traffic_src = round(rec["hbm_bytes_per_launch"]), os.path.basename(tfiles[-1])
            traffic_commit, traffic_digest = rec.get("commit"), rec.get("csrc_digest")
        if args.breakdown:
            rows = sorted(prof.by_tag().items(), key=lambda kv: -kv[1][2])
            for tag, (n, fl, t) in rows[:60]:
                sys.stderr.write("%-46s n=%3d  %8.3f ms  %7.2f TF/s\n" % (tag, n, t, fl / (t * 1e-3) / 1e12 if t > 0 else 0))
        steps_p = max(1, args.profile_steps)
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        result["roofline"] = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes/launch",
            "traffic_source": traffic_src,
            # the profile is a constant from the builder's box: which sources it was taken on, and whether they are still these
            "traffic_commit": traffic_commit, "traffic_csrc_digest": traffic_digest, "csrc_digest": csrc_digest(),
            "stale": None if traffic is None else (traffic_digest != csrc_digest()),
            "traffic_note": None if traffic is None else "constant read from the committed rocprofv3 PMC profile (separate --pmc passes on the "
                                                         "builder's box, tools/profile_round.sh), NOT measured during this run",
            "algorithmic_bytes_per_launch": round(prof.algo_bytes() / max(n_launch, 1)),
            "kernel": "gemm_f32_kernel (v_mfma_f32_32x32x2_f32 GEMM / implicit-GEMM conv fwd+dgrad+wgrad; 3x3 stride-1 "
                      "convolutions as Winograd F(4x4,3x3): their transform kernels are inside the timed span)",
            "launches_per_step": n_launch // steps_p,
            "algorithmic_gflop_per_step": round(flops / steps_p / 1e9, 1),
            # what the MFMA units execute: less than the algorithmic count where Winograd replaces the direct convolution
            "executed_gflop_per_step": round(executed / steps_p / 1e9, 1),
            "executed_tflops": round(executed / (ms * 1e-3) / 1e12, 2) if ms > 0 else 0.0,
            # frac counts algorithmic FLOPs; frac_executed what the MFMA units really execute (the number MFMA-busy counters track)
            "frac_executed": round(executed / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if ms > 0 else 0.0,
            "kernel_ms_per_step": round(ms / steps_p, 3),
            "whole_step_algorithmic_tflops": round(ALGO_GFLOP_PER_SAMPLE.get("image-only" if image_only else args.variant, 0) * B / ms_per_step, 2),
            "accounting": "achieved / frac count ALGORITHMIC FLOPs (SURVEY 8d: output pixels x taps x Cin x Cout of a convolution) over the "
                          "time of the launches that implement them; the Winograd F(4x4,3x3) path executes 4x fewer on the MFMA units "
                          "(executed_gflop_per_step, executed_tflops), so a convolution-only workload can exceed 1.0 here",
        }
        if args.dtype == "f32":
            # ONE meaning of achieved / frac on every fp32 line (vec32, img128, rad16): the FLOPs the MFMA units EXECUTE over the time
            # of the launches that implement the family (Winograd F(4x4,3x3) executes 1/4 of a 3x3 convolution's MACs; its transform
            # kernels are inside the timed span) - the quantity the MFMA-busy counters track and the conservative one.  SURVEY 8d's
            # ALGORITHMIC count (output pixels x taps x Cin x Cout) stays beside it as achieved_algorithmic / frac_algorithmic; on a
            # convolution-only workload (configs[3]) it exceeds 1.0.
            r = result["roofline"]
            r["frac_algorithmic"], r["achieved_algorithmic"] = r["frac"], r["achieved"]
            r["achieved"] = r["executed_tflops"]
            r["frac"] = r["frac_executed"]
            r["accounting"] = ("achieved / frac = FLOPs the MFMA units EXECUTE (executed_gflop_per_step) over the time of the family's launches "
                               "incl. the Winograd transform kernels, against the nominal fp32 MFMA peak; achieved_algorithmic / "
                               "frac_algorithmic count SURVEY 8d's algorithmic FLOPs (a 3x3 convolution as a direct convolution) over the same time")
            if not image_only and not args.no_attention_roofline:
                r["attention"] = attention_roofline(eng, B, dev)
        if args.dtype != "f32":
            # bf16 mode: the dominant kernel is the bf16-operand GEMM; price ITS launches against the dense bf16 MFMA peak.
            # What stays on the fp32 instruction (conv weight gradients in the Winograd domain, 7x7 stems, stride-2 data
            # gradients) is reported next to it against the fp32 peak.
            nb, fb, msb = prof.subset("bf16 ")
            r = result["roofline"]
            r["all_gemm_launches_vs_fp32_peak"] = {"achieved": r["achieved"], "peak": PEAK_FP32_MFMA_TFLOPS, "frac": r["frac"],
                                                   "launches_per_step": r["launches_per_step"], "kernel_ms_per_step": r["kernel_ms_per_step"]}
            ach = fb / (msb * 1e-3) / 1e12 if msb > 0 else 0.0
            r.update({"achieved": round(ach, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4),
                      "kernel": ("gemm16_nt / gemm16_tn kernels (v_mfma_f32_32x32x16_bf16, bf16 operands in HBM staged by global_load_lds, fp32 "
                                 "accumulate): Linear fwd / dX / dW and every trunk convolution fwd / data gradient / weight gradient; the 64- / 128-channel 3x3 "
                                 "stride-1 convolutions fwd / data gradient as conv16_halo_kernel (LDS-resident halo patch, the producer's BatchNorm in the loader)"
                                 if args.dtype == "bf16" else
                                 "gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16; fp32 operands in HBM rounded to bf16 on the way into LDS, "
                                 "fp32 accumulate): Linear GEMMs and direct 3x3 convolutions fwd / data gradient"),
                      "launches_per_step": nb // steps_p, "kernel_ms_per_step": round(msb / steps_p, 3),
                      "algorithmic_gflop_per_step": round(fb / steps_p / 1e9, 1), "traffic": None, "traffic_source": None,
                      "note": ("the step is HBM / launch-latency bound in this mode (SURVEY 8d): see hbm_roofline" if args.dtype == "bf16" else
                               "operands are read as fp32 from HBM (4 B/element), so these GEMMs are HBM/LDS-bound long before the "
                               "2.5 PFLOP/s bf16 MFMA peak; see DESIGN.md section 7")})
            if args.dtype == "bf16":
                # this mode is HBM / launch bound, not MFMA bound - the second roofline.  Measured bytes: rocprofv3
                # FETCH_SIZE (x2, the guide's gfx950 correction) + WRITE_SIZE over every kernel of a step (tools/profile_round.sh
                # with EXTRA="--dtype bf16"); algorithmic bytes: SURVEY 8d's activation estimate (317 MB/sample saved in fp32
                # -> write + read, halved for bf16) + parameter traffic (bf16 shadows read twice, their derivation, fp32
                # gradients written, AdamW 28 B/param)
                hb = sorted(glob.glob(os.path.join(ROOT, "profiles", "*bf16_traffic.json")))
                measured = json.load(open(hb[-1])).get("whole_step_hbm_bytes") if hb and default_like_bf16(args, B) else None
                algo = B * 317e6 * 2 / 2 + (2 * 210e6 + 419e6 + 382e6 + 419e6 + 2.93e9)
                r["hbm_roofline"] = {
                    "bound": "hbm", "peak": 8.0, "unit": "TB/s",
                    "algorithmic_bytes_per_step": round(algo), "achieved_algorithmic": round(algo / (ms_per_step * 1e-3) / 1e12, 3),
                    "frac_algorithmic": round(algo / (ms_per_step * 1e-3) / 8e12, 4),
                    "measured_bytes_per_step": None if measured is None else round(measured),
                    "achieved_measured": None if measured is None else round(measured / (ms_per_step * 1e-3) / 1e12, 3),
                    "frac_measured": None if measured is None else round(measured / (ms_per_step * 1e-3) / 8e12, 4),
                    "traffic_source": os.path.basename(hb[-1]) if (hb and measured is not None) else None}
                r["mode"] = dict(dtype_detail="bf16 activations / saved tensors / weight shadows in HBM; fp32 accumulation, BatchNorm and "
                                              "LayerNorm statistics, master weights, gradients, AdamW, loss head, VectorNet and the two 7x7 stems")
            for k in ("executed_gflop_per_step", "executed_tflops", "frac_executed", "algorithmic_bytes_per_launch"):
                r.pop(k, None)
        if not image_only and args.dtype in ("f32", "bf16") and not args.no_oracle_check:
            result["loss_vs_oracle"] = loss_vs_oracle(net, eng, inp, gt, args.variant)
            if args.dtype == "bf16":   # ~100 bf16 layers deep: the mode's stated tolerance is relative
                lv = result["loss_vs_oracle"]
                lv["tolerance"] = "2e-3 relative"
                lv["rel_diff"] = float("%.3g" % (abs(lv["hip"] - lv["oracle"]) / max(abs(lv["oracle"]), 1e-12)))
        if not args.no_cpu_baseline and not image_only and args.variant == "vec" and world == 1:   # rank 0 at N=1 only
            result["cpu_baseline"] = cpu_baseline()
        if world == 1 and default_workload and not args.no_also and args.config is None:
            # BASELINE configs[2] (per-GPU arithmetic), configs[3], configs[4]: driver-timed beside the headline
            also = {name: also_config(args, name) for name in ("bf16", "img128", "rad16")}
            # the other configurations' throughputs as top-level scalars ahead of the long blocks AND in a short last object: a
            # truncated copy of this line (head or tail) still carries them
            scal = {"also_" + k: (v or {}).get("value") for k, v in also.items()}
            head = {}
            for k, v in result.items():
                head[k] = v
                if k == "ms_per_step":
                    head.update(scal)
            result = head
            result["also"] = also
            result["tail"] = dict(value=result["value"], unit="samples/s", ms_per_step=result["ms_per_step"],
                                  roofline_frac=result.get("roofline", {}).get("frac"),
                                  roofline_frac_algorithmic=result.get("roofline", {}).get("frac_algorithmic"),
                                  attention_fwd_frac=result.get("roofline", {}).get("attention", {}).get("fwd", {}).get("frac"),
                                  attention_bwd_frac=result.get("roofline", {}).get("attention", {}).get("bwd", {}).get("frac"), **scal)
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
