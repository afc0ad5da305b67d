import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
dev = "cuda:0"
def t(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in ((6144, 2048, 512), (6144, 512, 2048), (6144, 1536, 512), (6144, 512, 512), (6144, 1024, 256), (6144, 256, 256), (16384, 4096, 4096)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev); dw = torch.empty(N, K, device=dev); dx = torch.empty(M, K, device=dev)
    ref = (x.double() @ w.double().t())
    y32 = ops.linear_fwd(x, w, out=torch.empty_like(y)); e32 = ((y32.double() - ref).abs().max() / ref.abs().max()).item()
    f32 = (t(lambda: ops.linear_fwd(x, w, out=y)), t(lambda: ops.linear_dx(dy, w, out=dx)), t(lambda: ops.linear_dw(dy, x, out=dw)))
    with ops.precision("f32x3"):
        y3 = ops.linear_fwd(x, w, out=torch.empty_like(y)); e3 = ((y3.double() - ref).abs().max() / ref.abs().max()).item()
        x3 = (t(lambda: ops.linear_fwd(x, w, out=y)), t(lambda: ops.linear_dx(dy, w, out=dx)), t(lambda: ops.linear_dw(dy, x, out=dw)))
    print((M, N, K), "fp32 MFMA fwd/dx/dw us: %.1f %.1f %.1f | bf16x3: %.1f %.1f %.1f | max err vs fp64 (rel to max): native %.1e, x3 %.1e" % (f32 + x3 + (e32, e3)))
