#!/bin/bash
# MFMA utilisation of the fusion-attention kernels at the bench shape (B = 32, T = 192, 4 heads, dropout 0.1 as in the step), all four
# head sizes, forward / dQ / dK-dV, fp32 (attention_wg.hip) and bf16 mode (attention16.hip):
#   pass 0: rocprofv3 --kernel-trace only -> durations without counters,
#   pass 1: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE (+ its own kernel trace).
# Usage (gpurun): bash tools/attn_profile.sh r05   -> gpurun_out/attn_r05/summary.txt
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/attn_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ATTN_DROP=0.1
for DT in f32 bf16; do
  ATTN_DTYPE=$DT timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t_$DT -- python $R/tools/attn_bench.py > $OUT/t_$DT.log 2>&1
  ATTN_DTYPE=$DT timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p_$DT -- python $R/tools/attn_bench.py > $OUT/p_$DT.log 2>&1
done
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
def short(k):
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    return k.split('(')[0]
print("# MFMA utilisation of the fusion-attention kernels, B = 32, T = 192, 4 heads, dropout 0.1 (tools/attn_profile.sh; isolated launches, 23 per kernel)")
print("#   busy      = SQ_VALU_MFMA_BUSY_CYCLES per launch (= 32 cycles x SQ_INSTS_MFMA for v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x16_bf16)")
print("#   util_pmc  = busy / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the counter pass's own clock and ITS (stretched) launch duration")
print("#   util_time = busy / (1024 SIMDs x launch duration WITHOUT counters x the shader clock of the counter pass), i.e. the same busy cycles over")
print("#               the time the launch takes when it is not being profiled (the counter pass stretches these 10-80 us launches by 1.2-2x)")
grand = {}
for dt in ("f32", "bf16"):
    dur0 = collections.defaultdict(list)
    for f in glob.glob(out + "/t_%s*kernel_trace.csv" % dt):
        for r in csv.DictReader(open(f)):
            dur0[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    dur1 = collections.defaultdict(list)
    for f in glob.glob(out + "/p_%s*kernel_trace.csv" % dt):
        for r in csv.DictReader(open(f)):
            dur1[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/p_%s*counter_collection.csv" % dt):
        for r in csv.DictReader(open(f)):
            cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("\n## %s" % ("fp32 path: attention_wg.hip (v_mfma_f32_16x16x4_f32)" if dt == "f32" else "bf16 mode: attention16.hip (v_mfma_f32_32x32x16_bf16)"))
    print("%-42s %6s %10s %9s %9s %8s %9s %9s %9s" % ("kernel", "n", "MFMA/launch", "us plain", "us pmc", "GHz", "util_pmc", "util_time", "wait_inst"))
    tb = tc = tp = 0.0
    rows = []
    for k, c in cnt.items():
        if "attn" not in k:
            continue
        med = lambda v: sorted(v)[len(v) // 2]
        busy, insts, gui = med(c["SQ_VALU_MFMA_BUSY_CYCLES"]), med(c["SQ_INSTS_MFMA"]), med(c["GRBM_GUI_ACTIVE"]) / 8.0
        wc, wi = med(c["SQ_WAVE_CYCLES"]), med(c["SQ_WAIT_INST_ANY"])
        d0, d1 = med(dur0[k]) if dur0.get(k) else float("nan"), med(dur1[k])
        ghz = gui / d1
        up, ut = busy / (gui * 1024), busy / (1024 * d0 * ghz)
        hs = int(re.search(r"<(\d+)", k).group(1))
        rows.append((hs, k, len(c["GRBM_GUI_ACTIVE"]), insts, d0 / 1e3, d1 / 1e3, ghz, up, ut, wi / wc))
        tb += busy; tc += gui * 1024; tp += 1024 * d0 * ghz
    for r in sorted(rows):
        print("%-42s %6d %10.0f %9.1f %9.1f %8.2f %9.3f %9.3f %9.3f" % r[1:])
    print("all %s attention kernels of a step, FLOP-weighted (sum of busy cycles / sum of capacity): util_pmc %.3f   util_time %.3f" % (dt, tb / tc, tb / tp))
PY
cat $OUT/summary.txt
