#!/bin/bash
# HBM traffic counters for the ingest / splat kernels: separate rocprofv3 passes (counters + kernel-trace only).
# FETCH_SIZE is doubled afterwards (MI355X_MICROARCH.md: on gfx950 it reports 1/2 of a wide coalesced read); both in KB.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ingest
rm -rf $OUT; mkdir -p $OUT
python $R/tools/ingest_bench.py > $OUT/timing.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o $C -- python $R/tools/ingest_bench.py --once > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
python - <<'PY' > $OUT/summary.txt
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/pmc_ingest'
print(open(out + '/timing.txt').read())
print("# rocprofv3 --kernel-trace --pmc <C> -- python tools/ingest_bench.py --once   (one launch per case, in program order)")
for f in sorted(glob.glob(out + '/**/*counter_collection.csv', recursive=True)):
    name = 'FETCH_SIZE' if 'FETCH' in f else 'WRITE_SIZE'
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'ingest' in k or 'splat' in k:
            v = float(r['Counter_Value'])
            mb = v * 1024 / 1e6 * (2 if name == 'FETCH_SIZE' else 1)
            print("%-12s %-60s grid %-10s raw %12.1f KB -> %8.2f MB%s" % (name, k.replace('(anonymous namespace)::', '')[:60], r['Grid_Size'], v, mb,
                  " (x2 gfx950 correction)" if name == 'FETCH_SIZE' else ""))
PY
cat $OUT/summary.txt
