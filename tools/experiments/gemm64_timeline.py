"""Per-k-tile issue timeline of the fp32 GEMM kernel from s_memtime stamps inside the kernel (an experiment build of
gemm_f32.hip with -DMMFN_GEMM_TIMELINE; no thread-trace decoder is installed in this image).  Run through
tools/experiments/gemm64_timeline.sh.  For every shape x tile: where a wave's time goes (until its first operands are there,
per k-tile, stores), how the blocks' starts and ends are spread, how many blocks / waves share a CU / SIMD."""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mmfn_amd import ops  # noqa: E402
from mmfn_amd._lib import lib  # noqa: E402

DEV = "cuda:0"
SLOTS = 48
L = lib()
L.mmfn_debug_gemm_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.mmfn_debug_gemm_timeline.restype = ctypes.c_int
WAVES = {1: 4, 2: 4, 3: 4, 4: 4, 5: 4, 6: 4, 7: 2}
DIMS = ops.TILE_DIMS


def pct(x, q):
    return float(np.percentile(x, q))


def run(name, fn, blocks, tile, nkt, mfma_per_ktile):
    waves = WAVES[tile]
    buf = torch.zeros(blocks * waves * SLOTS, dtype=torch.int64, device=DEV)
    L.mmfn_debug_gemm_timeline(None, 0)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L.mmfn_debug_gemm_timeline(buf.data_ptr(), blocks)
    fn()
    torch.cuda.synchronize()
    L.mmfn_debug_gemm_timeline(None, 0)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    t = buf.cpu().numpy().astype(np.uint64).reshape(blocks, waves, SLOTS)
    ok = t[:, :, 41] > 0
    if not ok.all():
        print("%s: %d of %d waves left no end stamp" % (name, int((~ok).sum()), ok.size))
    t = t[ok.all(axis=1)]
    st, en = t[:, :, 0].astype(np.float64), t[:, :, 41].astype(np.float64)
    rt0, rt1 = t[:, :, 43].astype(np.float64), t[:, :, 44].astype(np.float64)
    # ns per s_memtime tick (s_memrealtime counts at 100 MHz): calibrated on the longer half of the waves
    dur, rdur = (en - st).ravel(), (rt1 - rt0).ravel()
    sel = rdur >= np.percentile(rdur, 50)
    tick_ns = 10.0 * rdur[sel].sum() / max(dur[sel].sum(), 1.0)
    to_us = lambda x: x * tick_ns * 1e-3  # noqa: E731
    nk = min(nkt, 36)
    first = t[:, :, 3].astype(np.float64)
    kt = t[:, :, 3:3 + nk].astype(np.float64)
    dk = np.diff(kt, axis=2) if nk > 1 else np.zeros((t.shape[0], waves, 1))
    loop_end, pro = t[:, :, 40].astype(np.float64), t[:, :, 2].astype(np.float64)
    hw = t[:, :, 1].astype(np.int64)
    xcc = t[:, :, 42].astype(np.int64) & 15
    cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15))[:, 0]
    simd = (hw >> 4) & 3
    per_cu = collections.Counter(cu.tolist())
    per_simd = collections.Counter(zip(np.repeat(cu, waves).tolist(), simd.ravel().tolist()))
    mfma_ns = mfma_per_ktile * 64 * tick_ns   # 64 cycles per v_mfma_f32_32x32x2_f32 at the clock the stamps themselves measure
    print("== %s  tile %d (%dx%d, %d waves)  %d blocks, %d k-tiles, %.1f us per launch (events, 20 launches)" % (
        name, tile, DIMS[tile][0], DIMS[tile][1], waves, blocks, nkt, us))
    # (s_memtime counters of different CUs are not synchronised: only differences inside one wave are used)
    print("   s_memtime tick = %.2f ns = %.2f GHz shader clock; launch (events) minus the median wave's lifetime: %.1f us" % (
        tick_ns, 1.0 / tick_ns, us - to_us(np.median(en - st))))
    print("   per wave, us:  start -> prologue loads issued %.2f | -> first k-tile landed %.2f (p90 %.2f) | k-loop %.2f (p90 %.2f) | "
          "stores %.2f (p90 %.2f) | whole %.2f (p90 %.2f)" % (
              to_us(np.median(pro - st)), to_us(np.median(first - st)), to_us(pct(first - st, 90)),
              to_us(np.median(loop_end - first)), to_us(pct(loop_end - first, 90)),
              to_us(np.median(en - loop_end)), to_us(pct(en - loop_end, 90)), to_us(np.median(en - st)), to_us(pct(en - st, 90))))
    if nk > 1:
        d = dk.ravel() * tick_ns
        print("   k-tile interval, ns: p10 %.0f  p50 %.0f  p90 %.0f  mean %.0f   (its %d MFMAs alone: %.0f ns; x the waves sharing the SIMD, below)" % (
            pct(d, 10), pct(d, 50), pct(d, 90), d.mean(), mfma_per_ktile, mfma_ns))
        by_pos = dk.mean(axis=(0, 1)) * tick_ns
        print("   mean interval by k-tile index: " + " ".join("%.0f" % v for v in by_pos[:35]))
    print("   CUs used %d; blocks per CU: %s; waves per SIMD: %s" % (
        len(per_cu), dict(sorted(collections.Counter(per_cu.values()).items())), dict(sorted(collections.Counter(per_simd.values()).items()))))
    sys.stdout.flush()


def main():
    A_ROW, B_NK, B_KN = ops.A_ROWMAJOR, ops.B_NK, ops.B_KN
    tiles = [int(x) for x in os.environ.get("TL_TILES", "2,7").split(",")]
    # Winograd-domain forward / adjoint GEMMs of the four ResNet stages: 36 x [tiles x Cin] . [Cout x Cin]^T
    for (T, C, label) in ((512, 256, "layer3"), (2048, 128, "layer2"), (8192, 64, "layer1"), (128, 512, "layer4")):
        V = torch.randn(36, T, C, device=DEV)
        U = torch.randn(36, C, C, device=DEV)
        M = torch.empty(36, T, C, device=DEV)
        for form, bname in ((B_NK, "NT"), (B_KN, "NN")):
            for tile in tiles:
                bm, bn = DIMS[tile]
                blocks = 36 * (-(-T // bm)) * (-(-C // bn))
                per_wave_tiles = (bm // 32) * (bn // 32) // WAVES[tile]
                run("winograd %s %s 36 x %d x %d x %d" % (label, bname, T, C, C),
                    lambda: ops.gemm(V, U, M, T, C, C, C, C, C, A_ROW, form, batch=36, strideA=T * C, strideB=C * C, strideC=T * C,
                                     tile=tile, splitk=1),
                    blocks, tile, C // 16, 8 * per_wave_tiles)
    # transformer GEMMs (M = 6144 tokens)
    for (N, K, label, tl) in ((768, 256, "C=256 qkv", (2, 7, 4)), (256, 256, "C=256 proj", (2, 7)), (512, 512, "C=512 proj", (1, 3, 5)),
                              (2048, 512, "C=512 mlp.0", (1, 5))):
        x = torch.randn(6144, K, device=DEV)
        w = torch.randn(N, K, device=DEV)
        b = torch.randn(N, device=DEV)
        y = torch.empty(6144, N, device=DEV)
        for tile in tl:
            bm, bn = DIMS[tile]
            blocks = (-(-6144 // bm)) * (-(-N // bn))
            per_wave_tiles = (bm // 32) * (bn // 32) // WAVES[tile]
            run("linear fwd %s 6144 x %d x %d" % (label, N, K), lambda: ops.linear_fwd(x, w, b, out=y, tile=tile, splitk=1), blocks, tile,
                K // 16, 8 * per_wave_tiles)


if __name__ == "__main__":
    main()
