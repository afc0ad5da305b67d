"""Data-parallel step check, run under torch.distributed.run with 2+ ranks (tests/test_parallel_gpu.py drives it).

Backends
  * >= world devices visible: one rank per GPU over **RCCL** (`nccl`), the production path;
  * otherwise (the 1-GPU test box): all ranks share cuda:0 over gloo (RCCL refuses two ranks per device) - the same
    DataParallel / Engine / GraphedStep code on GPU tensors, only the transport differs.  MMFN_DIST_BACKEND overrides.

Checks
  1. rank lock-step: parameters and reduced gradients identical across ranks after the step;
  2. VALUE: the reduced gradient equals the mean of the per-rank CPU-oracle gradients (each rank runs the oracle on its own
     shard with per-rank BatchNorm statistics, exactly DDP's semantics, phase2_train_net.py:227,265-269), judged like
     tests/test_e2e_gpu.py against an fp64 oracle with the fp32 oracle's own error as yardstick; the updated weights equal
     a reference-style 2-rank loop (torch AdamW on the averaged oracle gradient) on every element whose sign is determined;
  3. the graph-replayed step (GraphedStep: collectives between hipGraph replays) reproduces the eager data-parallel step bit for bit.
"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def _oracle_grads(model, args, gt):
    from oracle import harness
    model.train()
    for p in model.parameters():
        p.grad = None
    loss = harness.l1_waypoint_loss(model(*args), gt)
    loss.backward()
    return loss.detach(), {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}


def main():
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    backend = os.environ.get("MMFN_DIST_BACKEND") or ("nccl" if ndev >= world else "gloo")
    local = int(os.environ.get("LOCAL_RANK", rank)) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
        if rank == 0:
            print("NOTE: %d device(s) visible for %d ranks -> ranks share cuda:0 over gloo; the RCCL transport is NOT exercised"
                  % (ndev, world), flush=True)
    import bench
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    from mmfn_amd.parallel import DataParallel, GraphedStep
    from oracle import harness
    torch.set_num_threads(max(1, bench.usable_cores() // world))
    torch.manual_seed(100 + rank)               # deliberately different init: broadcast must fix it
    net = MMFN(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0), dev); net.train()
    oracle = harness.build_oracle("vec", dropout=0.0)   # closed-form weights: identical on every rank
    if rank == 0:
        net.load_state_dict(oracle.state_dict(), strict=True)
    comm = None
    if os.environ.get("MMFN_DP_TRANSPORT", "auto") != "torch" and backend == "nccl":
        # buckets through libmmfn_comm.so (C ABI over RCCL; the step is then ONE hipGraph with the collectives inside) - the
        # transport bench.py selects by itself when the library's communicator comes up
        from mmfn_amd.comm import open_transport
        comm, _ = open_transport(rank, world, dist, dev, required=os.environ.get("MMFN_DP_TRANSPORT") == "capi")
    dp = DataParallel(net, dist, comm=comm); dp.broadcast_parameters()
    batch = int(os.environ.get("DP_CHECK_BATCH", "2"))   # per rank; the nccl test runs the benched 32
    inp, gt = bench.synth_inputs(batch, dev, seed=7 + rank, lanes=16 if batch <= 2 else 64, n_lidar=4096 if batch <= 2 else 16384)
    L = net._layout
    eng = net._engine_for()
    p0 = L.params.clone()
    snap = (L.params.clone(), L.exp_avg.clone(), L.exp_avg_sq.clone(), L.buffers_flat.clone(), L.counters_flat.clone(),
            eng.step_count.clone(), eng.rng_state.clone())

    def restore():
        for dst, src in zip((L.params, L.exp_avg, L.exp_avg_sq, L.buffers_flat, L.counters_flat, eng.step_count, eng.rng_state), snap):
            dst.copy_(src)

    # ---- 1. one eager data-parallel step
    loss = net.train_step(inp, gt, dp=dp)
    torch.cuda.synchronize()
    ps = [torch.empty_like(L.params) for _ in range(world)]
    dist.all_gather(ps, L.params)
    same = all(torch.equal(ps[0], p) for p in ps)
    gs = [torch.empty_like(L.grads) for _ in range(world)]
    dist.all_gather(gs, L.grads)
    same_g = all(torch.equal(gs[0][:L.tail], g[:L.tail]) for g in gs)
    moved = (L.params[:L.tail] - p0[:L.tail]).abs().max().item()

    # ---- 2. value: mean over ranks of the oracle's per-rank gradients
    args = harness.forward_args(bench.oracle_batch_from_inputs(inp, "vec"), "vec")
    o64 = copy.deepcopy(oracle).double()
    to64 = lambda a: (a.double() if torch.is_tensor(a) and a.is_floating_point() else
                      (type(a)(to64(x) for x in a) if isinstance(a, (list, tuple)) else a))
    loss32, g32 = _oracle_grads(oracle, args, gt.cpu())
    _, g64 = _oracle_grads(o64, to64(args), gt.cpu().double())
    cpu_pg = dist.new_group(backend="gloo") if backend == "nccl" else None   # CPU tensors need a gloo group

    def mean_over_ranks(t):
        t = t.clone()
        dist.all_reduce(t, group=cpu_pg)
        return t / world

    net._layout.attach_grads()
    bad, ratios, checked, total = [], [], 0, 0
    named = dict(net.named_parameters())
    ref_params = dict(oracle.named_parameters())
    gmax = None
    means = {}
    for name in sorted(g64):
        if g64[name] is None:
            continue
        means[name] = (mean_over_ranks(g32[name].double()), mean_over_ranks(g64[name]))
    gmax = max(m64.norm().item() for _, m64 in means.values())
    rel_cpu = sorted((m32 - m64).norm().item() / m64.norm().item() for m32, m64 in means.values() if m64.norm().item() > 1e-6 * gmax)
    med_cpu = rel_cpu[len(rel_cpu) // 2]   # typical fp32-oracle error: the yardstick where the oracle happened to land unusually close
    for name, (m32, m64) in means.items():
        hip = named[name].grad.detach().cpu().double() / world        # flat buffer holds the SUM; 1/world is folded into AdamW
        n = m64.norm().item()
        e_hip, e_cpu = (hip - m64).norm().item(), (m32 - m64).norm().item()
        if n > 1e-6 * gmax:
            ratios.append(e_hip / max(e_cpu, 1e-12 * gmax))
        if e_hip > 12.0 * max(e_cpu, med_cpu * n) + 2e-4 * n + 1e-8 * gmax:
            bad.append((name, e_hip, e_cpu, n))
        ref_params[name].grad = m32.float()
    ratios.sort()
    # reference-style loop on the averaged gradient: torch AdamW on the oracle's parameters
    init = {k: v.detach().clone() for k, v in oracle.named_parameters()}
    torch.optim.AdamW(oracle.parameters(), lr=1e-4).step()
    for name, (m32, m64) in means.items():
        ghip = named[name].grad.detach().cpu().double() / world
        sure = m64.abs() > 10.0 * torch.maximum((m32 - m64).abs(), (ghip - m64).abs()) + 1e-7 * gmax
        upd_ref = ref_params[name].detach().double() - init[name].double()
        upd_hip = named[name].detach().cpu().double() - init[name].double()
        total += m64.numel(); checked += int(sure.sum())
        if sure.any() and (upd_hip - upd_ref)[sure].abs().max().item() > 2e-6:
            bad.append((name, "update", (upd_hip - upd_ref)[sure].abs().max().item()))
    loss_ok = abs(loss.item() - loss32.item()) <= 1e-4
    value_ok = (not bad) and ratios[len(ratios) // 2] <= 2.5 and checked >= 0.1 * total and loss_ok

    # ---- 3. the graph-replayed step == the eager data-parallel step, bit for bit
    steps = int(os.environ.get("DP_CHECK_STEPS", "1"))
    restore()
    for _ in range(steps):
        net.train_step(inp, gt, dp=dp)
    torch.cuda.synchronize()
    eager = L.params.clone()
    restore()
    seg = GraphedStep(eng, dp, inp, gt, lr=1e-4, warm=0)
    restore()
    dp.measure_exposed = True
    for _ in range(steps):
        seg()
    torch.cuda.synchronize()
    same_seg = torch.equal(eager, L.params)
    exposed = dp.exposed_ms()
    flags = torch.tensor([1.0 if (same_seg and value_ok and same and same_g) else 0.0])
    dist.all_reduce(flags, group=cpu_pg)
    if rank == 0:
        print("backend:", backend, "| ranks:", world, "| devices:", ndev, "| gradient transport:",
              "C ABI (mmfn_allreduce_sum_f32)" if comm is not None else "torch.distributed", "| batch/rank:", batch,
              "| buckets:", dp.n_buckets(), "| graphs per step:", seg.recorder.n_graphs)
        print("params identical across ranks:", same, "| reduced grads identical:", same_g, "| max |dp|: %.3e" % moved,
              "| loss %.6f (oracle %.6f)" % (loss.item(), loss32.item()), "| segmented graphs == eager:", same_seg)
        print("reduced gradient == mean of per-rank oracle gradients:", value_ok, "| median error ratio %.2f" % ratios[len(ratios) // 2],
              "| update-checked elements %.0f%%" % (100.0 * checked / max(total, 1)), "| exposed comm %.3f ms/step" % (exposed or 0.0))
        if bad:
            print("BAD:", bad[:6])
        sys.stdout.flush()
    if rank == 0:
        assert same and same_g and 0 < moved < 1e-3 and same_seg and value_ok
    assert flags.item() == world, "a rank failed the data-parallel check"
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
