"""Build and time tools/experiments/bf16_gemm.hip against the fp32 MFMA GEMM of the library (same shapes)."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from mmfn_amd import ops
so = os.path.join(HERE, "bf16_gemm.so")
if "--build" in sys.argv or not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", os.path.join(HERE, "bf16_gemm.hip"), "-o", so])
    if "--build" in sys.argv: sys.exit(0)
lib = ctypes.CDLL(so)
dev = "cuda:0"
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in ((6144, 2048, 512), (6144, 512, 2048), (6144, 1536, 512), (6144, 512, 512), (16384, 4096, 4096), (73728, 256, 256)):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev); C32 = torch.empty(M, N, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: lib.exp_gemm_bf16_nt(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B.data_ptr()), ctypes.c_void_p(C.data_ptr()), M, N, K, ctypes.c_void_p(st))
    assert run() == 0
    ref = A.bfloat16().float() @ B.bfloat16().float().t()
    err = (C - ref).abs().max().item() / ref.abs().max().item()
    us16 = t(run); us32 = t(lambda: ops.linear_fwd(A, B, out=C32))
    fl = 2.0 * M * N * K
    print("%6dx%5dx%5d: bf16-operand %7.1f us %6.1f TF/s | fp32 MFMA %7.1f us %6.1f TF/s | x%.2f | max rel err vs bf16-rounded fp32 matmul %.1e" % (
        M, N, K, us16, fl / us16 / 1e6, us32, fl / us32 / 1e6, us32 / us16, err))
