"""What a cross-branch grouped launch could buy (round-5 review item 4): the camera and the map branch run the same ResNet-34
shapes, so their Winograd-domain GEMMs could be ONE batch-72 launch instead of two batch-36 launches on two streams.  Per trunk
shape and GEMM form: two batch-36 launches back to back on one stream, the same two on two streams (what the lanes do today), and
one batch-72 launch over both branches' operands (contiguous here; a pointer table in a real implementation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmfn_amd import ops
dev = "cuda:0"
iters = 30


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print("%-28s %12s %12s %12s %10s" % ("form / shape (B = 32)", "2 x 36 serial", "2 x 36 two streams", "1 x 72", "72 vs streams"))
tot = [0.0, 0.0, 0.0]
for (T, C, n_convs) in ((8192, 64, 6), (2048, 128, 7), (512, 256, 11), (128, 512, 5)):
    V = torch.randn(72, T, C, device=dev); U = torch.randn(72, C, C, device=dev); M = torch.empty(72, T, C, device=dev)
    dU = torch.empty(72, C, C, device=dev)
    forms = {
        "fwd NT": lambda nb, o: ops.gemm(V[o:], U[o:], M[o:], T, C, C, C, C, C, ops.A_ROWMAJOR, ops.B_NK, batch=nb, strideA=T * C, strideB=C * C, strideC=T * C),
        "adj NN": lambda nb, o: ops.gemm(V[o:], U[o:], M[o:], T, C, C, C, C, C, ops.A_ROWMAJOR, ops.B_KN, batch=nb, strideA=T * C, strideB=C * C, strideC=T * C),
        "wgrad TN": lambda nb, o: ops.gemm(M[o:], V[o:], dU[o:], C, C, T, C, C, C, ops.A_COLMAJOR, ops.B_KN, batch=nb, strideA=T * C, strideB=T * C, strideC=C * C),
    }
    for name, f in forms.items():
        serial = timeit(lambda: (f(36, 0), f(36, 36)))

        def two():
            ev = torch.cuda.Event(); ev.record()
            s1.wait_event(ev); s2.wait_event(ev)
            with torch.cuda.stream(s1), ops.lane(1):
                f(36, 0)
            with torch.cuda.stream(s2), ops.lane(2):
                f(36, 36)
            e1, e2 = torch.cuda.Event(), torch.cuda.Event()
            e1.record(s1); e2.record(s2)
            torch.cuda.current_stream().wait_event(e1); torch.cuda.current_stream().wait_event(e2)
        par = timeit(two)
        one = timeit(lambda: f(72, 0))
        print("%-28s %12.1f %12.1f %12.1f %9.1f us" % ("%s T=%d C=%d" % (name, T, C), serial, par, one, par - one))
        tot[0] += serial * n_convs; tot[1] += par * n_convs; tot[2] += one * n_convs
print("summed over the 29 same-shape 3x3 convolutions of layers 1-4 (camera + map), us per step: serial %.0f, two streams %.0f, grouped %.0f" % tuple(tot))
