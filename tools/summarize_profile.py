"""Condense rocprofv3 CSV output into the small text summaries committed under profiles/."""
import collections
import csv
import glob
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:80]


# 1. kernel stats
for f in glob.glob(os.path.join(out, "*kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, "%s_kernel_stats.txt" % tag), "w") as w:
        w.write("# rocprofv3 --kernel-trace --stats : python bench.py --no-graph --steps 5 --warmup 2 (10 train steps in total)\n")
        w.write("%-82s %8s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in rows[:45]:
            w.write("%-82s %8s %14.1f %12.2f %7.2f\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                      float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    print(open(os.path.join(dst, "%s_kernel_stats.txt" % tag)).read()[:3000])

# 2. PMC summaries per kernel name
def pmc(fname):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(fname)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            calls[k] += 1
    return agg, calls


lines = []
f = glob.glob(os.path.join(out, "pmc_mfma*counter_collection.csv"))
if f:
    agg, calls = pmc(f[0])
    lines.append("# MFMA utilisation per kernel family: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)")
    lines.append("%-70s %7s %14s %14s %9s %9s" % ("kernel", "calls", "mfma_busy_cyc", "gui_active", "mfma_util", "wait_inst"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:14]:
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        util = busy / (gui / 8.0 * 1024.0) if gui else 0.0
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        wi = v.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else 0.0
        lines.append("%-70s %7d %14.0f %14.0f %9.3f %9.3f" % (k, calls[k], busy, gui, util, wi))
for name, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(out, name + "*counter_collection.csv"))
    if f:
        agg, calls = pmc(f[0])
        lines.append("")
        lines.append("# %s (rocprofv3 raw units, KiB) summed over all launches, and per launch" % cname)
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(cname, 0))[:12]:
            lines.append("%-70s calls %6d  total %14.0f  per_launch %12.1f" % (k, calls[k], v[cname], v[cname] / max(calls[k], 1)))
open(os.path.join(dst, "%s_pmc.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))


# 3. HBM traffic of the dominant (GEMM) kernel family, per launch, for bench.py's roofline.traffic
import json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import csrc_digest  # noqa: E402
fetch = glob.glob(os.path.join(out, "pmc_fetch*counter_collection.csv"))
write = glob.glob(os.path.join(out, "pmc_write*counter_collection.csv"))
if fetch and write:
    FAMILY = os.environ.get("PROFILE_FAMILY", "gemm_f32")   # "gemm16" for the bf16 mode

    def fam(fname, cname, every=False):
        tot, ids = 0.0, set()
        for r in csv.DictReader(open(fname)):
            if (every or FAMILY in r["Kernel_Name"]) and r["Counter_Name"] == cname:
                tot += float(r["Counter_Value"])
                ids.add(r["Dispatch_Id"])
        return tot, len(ids)
    f, nf = fam(fetch[0], "FETCH_SIZE")
    w, nw = fam(write[0], "WRITE_SIZE")
    fa, _ = fam(fetch[0], "FETCH_SIZE", True)
    wa, _ = fam(write[0], "WRITE_SIZE", True)
    STEPS = 4   # the SHORT command: 2 sizing steps + 1 timed + 1 instrumented
    rec = {
        "kernel_family": FAMILY + "*", "launches_profiled": nf,
        # every kernel of the run (model construction included: a few hundred MB once), per train step
        "whole_step_hbm_bytes": (2.0 * fa + wa) * 1024.0 / STEPS, "whole_step_fetch_bytes_corrected": 2.0 * fa * 1024.0 / STEPS,
        "whole_step_write_bytes": wa * 1024.0 / STEPS, "steps_profiled": STEPS,
        "fetch_bytes_per_launch_raw": f * 1024.0 / max(nf, 1),
        "fetch_bytes_per_launch_corrected": 2.0 * f * 1024.0 / max(nf, 1),
        "write_bytes_per_launch": w * 1024.0 / max(nw, 1),
        "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0 / max(nf, 1),
        # which kernel sources the profile is valid for (bench.py marks the constant `stale` when its own digest differs); the
        # commit is stamped by tools/stamp_profiles.py when the file is copied into profiles/ (the GPU box has no .git)
        "csrc_digest": csrc_digest(), "commit": os.environ.get("MMFN_COMMIT"),
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                  "`bench.py --no-graph --steps 1 --warmup 0 --profile-steps 1`; counters are KiB; FETCH_SIZE doubled per "
                  "MI355X_MICROARCH.md (gfx950 reports half the bytes of 16-B/lane streaming reads); WRITE_SIZE matches "
                  "M*N*4 exactly on the isolated GEMM (tools/pmc_traffic.sh)",
    }
    json.dump(rec, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
    print(rec)
