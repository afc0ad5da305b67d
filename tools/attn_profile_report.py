#!/usr/bin/env python
"""profiles/rNN_attn_pmc.txt from the rocprofv3 output of tools/attn_profile.sh (gpurun_out/attn_<tag>/): per kernel the MFMA
instructions per launch, the launch duration with and without counters, and the MFMA utilisation three ways - by GRBM_GUI_ACTIVE (the
normalisation of profiles/rNN_pmc.txt), and over the kernel's own un-profiled duration at the nominal 2.4 GHz and at the 2.0 GHz the
chip sustains under fp32 MFMA load.

    python tools/attn_profile_report.py gpurun_out/attn_r05 > profiles/r05_attn_pmc.txt
"""
import collections
import csv
import glob
import re
import sys

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/attn_r05"


def short(k):
    return k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


print("""# MFMA utilisation of the fusion-attention kernels (north_star: >= 40 %)
# B = 32, T = 192, 4 heads, dropout 0.1 as in the training step; isolated launches, 23 per kernel; gpurun -- bash tools/attn_profile.sh <tag>,
# then python tools/attn_profile_report.py gpurun_out/attn_<tag>
#   us plain   launch duration under rocprofv3 --kernel-trace alone;  us pmc: under the counter pass
#   busy       SQ_VALU_MFMA_BUSY_CYCLES per launch = 32 cycles x SQ_INSTS_MFMA (v_mfma_f32_16x16x4_f32; v_mfma_f32_32x32x16_bf16)
#   util_pmc   busy / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) - the normalisation of profiles/r0x_pmc.txt.  GRBM_GUI_ACTIVE / 8 divided by the
#              launch's own duration in the SAME pass is 2.7-3.8 "GHz", above the 2.4 GHz maximum clock: the counter's window contains
#              7-13 us per launch that are not kernel time (dispatch, cache maintenance between launches), so util_pmc understates what
#              the matrix pipes do WHILE THE KERNEL RUNS - by a third for a 35 us launch, by half for a 15 us one
#   util@2.4   busy / (1024 SIMDs x us plain x 2.4 GHz): kernel-time utilisation against the NOMINAL peak (what 157.3 TF/s assumes)
#   util@2.0   the same at the 2.0 GHz the chip sustains under fp32 MFMA load (profiles/r05_gemm64_att.txt: 1.95-2.06 GHz by s_memtime)""")
for dt in ("f32", "bf16"):
    dur0, dur1 = collections.defaultdict(list), collections.defaultdict(list)
    for tag, d in (("t", dur0), ("p", dur1)):
        for f in glob.glob(out + "/%s_%s*kernel_trace.csv" % (tag, dt)):
            for r in csv.DictReader(open(f)):
                d[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/p_%s*counter_collection.csv" % dt):
        for r in csv.DictReader(open(f)):
            cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("\n## %s" % ("fp32 path: attention_wg.hip (v_mfma_f32_16x16x4_f32, peak 157.3 TF/s)" if dt == "f32"
                       else "bf16 mode: attention16.hip (v_mfma_f32_32x32x16_bf16, peak 2.5 PF/s)"))
    print("%-40s %11s %9s %8s %9s %9s %9s %9s" % ("kernel", "MFMA/launch", "us plain", "us pmc", "util_pmc", "util@2.4", "util@2.0", "wait_inst"))
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
    rows, tot = [], collections.defaultdict(float)
    for k, c in cnt.items():
        if "attn" not in k:
            continue
        busy, insts, gui = med(c["SQ_VALU_MFMA_BUSY_CYCLES"]), med(c["SQ_INSTS_MFMA"]), med(c["GRBM_GUI_ACTIVE"]) / 8.0
        d0, d1 = med(dur0[k]), med(dur1[k])
        hs = int(re.search(r"<(\d+)", k).group(1))
        rows.append((hs, k, insts, d0 / 1e3, d1 / 1e3, busy / (gui * 1024), busy / (1024 * d0 * 2.4), busy / (1024 * d0 * 2.0),
                     med(c["SQ_WAIT_INST_ANY"]) / med(c["SQ_WAVE_CYCLES"])))
        for key, v in (("busy", busy), ("gui", gui * 1024), ("c24", 1024 * d0 * 2.4), ("c20", 1024 * d0 * 2.0)):
            tot[key] += v
            tot[(hs, key)] += v
    for r in sorted(rows):
        print("%-40s %11.0f %9.1f %8.1f %9.3f %9.3f %9.3f %9.3f" % r[1:])
    for hs in (16, 32, 64, 128):
        print("   head size %3d, forward + backward:            util_pmc %.3f   util@2.4 %.3f   util@2.0 %.3f" % (
            hs, tot[(hs, "busy")] / tot[(hs, "gui")], tot[(hs, "busy")] / tot[(hs, "c24")], tot[(hs, "busy")] / tot[(hs, "c20")]))
    print("   all four transformers, FLOP-weighted (sum busy / sum capacity): util_pmc %.3f   util@2.4 %.3f   util@2.0 %.3f" % (
        tot["busy"] / tot["gui"], tot["busy"] / tot["c24"], tot["busy"] / tot["c20"]))
