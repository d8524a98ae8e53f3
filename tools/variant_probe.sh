#!/bin/bash
# A/B of library variants (build/variants/<name>.so) on the headline step: per-kernel average durations (rocprofv3 --stats) and
# HBM write bytes (PMC WRITE_SIZE pass).   bash tools/variant_probe.sh default sddmm_nt0 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/variant_probe
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset TSGU_LIB_PATH; else export TSGU_LIB_PATH=$ROOT/build/variants/$v.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v/stats -o s -- python3 $ROOT/tools/pattern_steps.py headline 200 > $OUT/$v.stats.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "tsgu::march_kernel" --output-format csv -d $OUT/$v/write -o p -- python3 $ROOT/tools/pattern_steps.py headline 6 > $OUT/$v.write.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "tsgu::march_kernel" --output-format csv -d $OUT/$v/fetch -o p -- python3 $ROOT/tools/pattern_steps.py headline 6 > $OUT/$v.fetch.log 2>&1
  echo "=== $v"
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$v/stats/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "march_kernel" in r["Name"]:
        print("  %-70s calls %5s avg %8.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
for what in ("write", "fetch"):
    f = glob.glob("$OUT/$v/%s/**/*counter_collection.csv" % what, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, vals in acc.items():
        tail = vals[len(vals) // 2:]
        print("  %-5s %-60s %10.1f KB per launch" % (what, k, sum(tail) / len(tail)))
PY
done 2>&1 | tee $OUT/summary.txt
