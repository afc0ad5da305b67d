"""Phase time stamps (s_memtime) of workgroup 0 of the fp32 attention forward kernel, waves 0 and 7: experiment build of
attention_wg.hip with -DMMFN_ATTN_STAMPS (tools/experiments/attn_phases.sh)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmfn_amd import ops
from mmfn_amd._lib import lib
dev = "cuda:0"
B, T, NH = 32, 192, 4
rng = torch.tensor([5, 1], dtype=torch.int64, device=dev)
buf = (ctypes.c_int64 * 32)()
for P in (0.0, 0.1):
    for HS in (16, 32, 64, 128):
        C = NH * HS
        qkv = torch.randn(B * T, 3 * C, device=dev)
        o = torch.empty(B * T, C, device=dev); lse = torch.empty(B, NH, T, device=dev)
        fwd = lambda: ops.attention_fwd(qkv[:, C:], qkv, qkv[:, 2 * C:], 3 * C, o, C, lse, B, T, NH, HS, 1 / math.sqrt(HS), drop_p=P, rng_state=rng, rng_stream=3)
        for _ in range(5): fwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20): fwd()
        e1.record(); torch.cuda.synchronize()
        lib().mmfn_attn_debug_read(ctypes.cast(buf, ctypes.c_void_p))
        for w, off in ((0, 0), (7, 16)):
            t = [buf[off + i] for i in range(6)]
            print("drop %.1f HS=%3d wave %d: launch %.1f us (events) | cycles: QK^T (operands + MFMA) %6d | softmax + V staged %6d | PV %6d | merge + store %6d | total %6d" % (
                P, HS, w, e0.elapsed_time(e1) / 20 * 1e3, t[2] - t[0], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]))
