import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
dev = torch.device("cuda:0")
side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
x = [torch.zeros(1 << 20, device=dev) for _ in range(3)]

def branches(fns, use_wait_stream=False):
    main = torch.cuda.current_stream()
    if use_wait_stream:
        for st in side: st.wait_stream(main)
    else:
        fork = torch.cuda.Event(); fork.record(main)
        for st in side: st.wait_event(fork)
    for i, f in enumerate(fns[1:]):
        with torch.cuda.stream(side[i]):
            f()
    fns[0]()
    if use_wait_stream:
        for st in side: main.wait_stream(st)
    else:
        for st in side:
            done = torch.cuda.Event(); done.record(st); main.wait_event(done)

def work():
    branches([lambda i=i: ops.fill(x[i], float(i + 1)) for i in range(3)], mode)
    ops.axpby(x[0], x[1], 1.0, 1.0)

for mode in (True, False):
    work(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            work()
        g.replay(); torch.cuda.synchronize()
        print("mode wait_stream=%s OK" % mode, x[0][0].item())
    except Exception as e:
        print("mode wait_stream=%s FAILED: %s" % (mode, str(e).split("\n")[0]))
