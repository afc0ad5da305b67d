"""Dev/CI tool: 2 ranks on ONE GPU over gloo (RCCL refuses two ranks per device) — exercises the real
DataParallel + Engine.train_step overlap path on GPU tensors and checks that ranks stay in lock-step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

def main():
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    import bench
    from mmfn_amd.config import GlobalConfig
    from mmfn_amd.model import MMFN
    from mmfn_amd.parallel import DataParallel
    torch.manual_seed(100 + rank)               # deliberately different init: broadcast must fix it
    net = MMFN(GlobalConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0), dev); net.train()
    dp = DataParallel(net, dist); dp.broadcast_parameters()
    inp, gt = bench.synth_inputs(2, dev, seed=7 + rank, lanes=16, n_lidar=4096)
    L = net._layout
    p0 = L.params.clone()
    steps = int(os.environ.get("DP_CHECK_STEPS", "1"))
    for _ in range(steps):
        loss = net.train_step(inp, gt, dp=dp)
    torch.cuda.synchronize()
    ps = [torch.empty_like(L.params) for _ in range(world)]
    dist.all_gather(ps, L.params)
    same = all(torch.equal(ps[0], p) for p in ps)
    gs = [torch.empty_like(L.grads) for _ in range(world)]
    dist.all_gather(gs, L.grads)
    same_g = all(torch.equal(gs[0][:L.tail], g[:L.tail]) for g in gs)
    moved = (L.params[:L.tail] - p0[:L.tail]).abs().max().item()
    # the five-graph step (collectives between hipGraph replays) must reproduce the eager data-parallel step bit for bit
    from mmfn_amd.parallel import GraphedStep
    eng = net._engine_for()
    snap = (L.params.clone(), L.exp_avg.clone(), L.exp_avg_sq.clone(), L.buffers_flat.clone(), L.counters_flat.clone(),
            eng.step_count.clone(), eng.rng_state.clone())

    def restore():
        for dst, src in zip((L.params, L.exp_avg, L.exp_avg_sq, L.buffers_flat, L.counters_flat, eng.step_count, eng.rng_state), snap):
            dst.copy_(src)

    for _ in range(steps):
        net.train_step(inp, gt, dp=dp)
    torch.cuda.synchronize()
    eager = L.params.clone()
    restore()
    seg = GraphedStep(eng, dp, inp, gt, lr=1e-4, warm=0)
    restore()
    for _ in range(steps):
        seg()
    torch.cuda.synchronize()
    same_seg = torch.equal(eager, L.params)
    if rank == 0:
        print("params identical across ranks:", same, "| reduced grads identical:", same_g, "| max |dp|: %.3e" % moved,
              "| loss %.5f" % loss.item(), "| segmented graphs == eager:", same_seg)
        assert same and same_g and 0 < moved < 1e-3 and same_seg
    ok = torch.tensor([1.0 if same_seg else 0.0])
    dist.all_reduce(ok)
    assert ok.item() == world, "segmented-graph step diverged from the eager step on some rank"
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    main()
