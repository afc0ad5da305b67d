#!/bin/bash
# For every BASELINE config a number is quoted for: the bench line (hipGraph replay) and a rocprofv3 --kernel-trace --stats
# summary of the same workload (eager, single stream, 3 steps) -> gpurun_out/prof_cfg/<tag>_<name>.{json,txt}
TAG=${1:-r02b}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_cfg
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  python $R/bench.py --no-cpu-baseline --no-oracle-check "$@" > $OUT/${TAG}_$name.json 2> $OUT/$name.err
  rm -rf $OUT/tr_$name
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$name -o t -- python $R/bench.py --no-graph --single-stream --no-cpu-baseline --no-oracle-check --steps 2 --warmup 1 --profile-steps 1 "$@" > $OUT/tr_$name.log 2>&1
  python3 - "$OUT/tr_$name" "$OUT/${TAG}_${name}_kernel_stats.txt" "$name: bench.py $*" <<'PY'
import csv, glob, sys
src, dst, title = sys.argv[1:4]
f = glob.glob(src + "/*kernel_stats.csv")
rows = list(csv.DictReader(open(f[0]))) if f else []
with open(dst, "w") as w:
    w.write("# rocprofv3 --kernel-trace --stats, eager single-stream launches, 4 train steps in total; %s\n" % title)
    w.write("%-82s %8s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows[:30]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
        w.write("%-82s %8s %14.1f %12.2f %7.2f\n" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  rm -rf $OUT/tr_$name
  echo "$name: $(cut -c1-200 $OUT/${TAG}_$name.json)"
}
run rad_b16_65536 --config rad16
run img_b32 --variant img
run image_only_b128 --config img128
run bf16_b32 --config bf16
run bf16_img_b32 --dtype bf16 --variant img
run bf16_rad_b16_65536 --config rad16 --dtype bf16
run bf16_image_only_b128 --dtype bf16 --config img128
run bf16_operands_b32 --dtype bf16-operands
run vec_19x8 --lane-format 19x8
MMFN_F32X3=1 run vec_f32x3
