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
    inp, gt = bench.synth_inputs(4, dev, seed=7 + rank)
    L = net._layout
    p0 = L.params.clone()
    for _ in range(2):
        loss = net.train_step(inp, gt, dp=dp)
    torch.cuda.synchronize()
    ps = [torch.empty_like(L.params) for _ in range(world)]
    dist.all_gather(ps, L.params)
    same = all(torch.equal(ps[0], p) for p in ps)
    gs = [torch.empty_like(L.grads) for _ in range(world)]
    dist.all_gather(gs, L.grads)
    same_g = all(torch.equal(gs[0][:L.tail], g[:L.tail]) for g in gs)
    moved = (L.params[:L.tail] - p0[:L.tail]).abs().max().item()
    if rank == 0:
        print("params identical across ranks:", same, "| reduced grads identical:", same_g, "| max |dp|: %.3e" % moved,
              "| loss %.5f" % loss.item())
        assert same and same_g and 0 < moved < 1e-3
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    main()
