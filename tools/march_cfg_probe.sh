#!/bin/bash
# per-kernel durations and HBM write / fetch bytes of the C2 step under TSGU_MARCH_CFG overrides:  bash tools/march_cfg_probe.sh "4,8,3,256" "2,16,3,256" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/march_cfg_probe
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for cfg in "$@"; do
  tag=$(echo $cfg | tr ',' '_')
  if [ "$cfg" = default ]; then unset TSGU_MARCH_CFG; else export TSGU_MARCH_CFG=$cfg; fi
  rm -rf $OUT/$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/stats -o s -- python3 $ROOT/tools/pattern_steps.py headline 100 > $OUT/$tag.stats.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "tsgu::march_kernel" --output-format csv -d $OUT/$tag/write -o p -- python3 $ROOT/tools/pattern_steps.py headline 6 > $OUT/$tag.write.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "tsgu::march_kernel" --output-format csv -d $OUT/$tag/fetch -o p -- python3 $ROOT/tools/pattern_steps.py headline 6 > $OUT/$tag.fetch.log 2>&1
  echo "=== $cfg"
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$tag/stats/**/*kernel_stats.csv", recursive=True)
if not f:
    print("  no stats (configuration rejected?)"); raise SystemExit
t = {}
for r in csv.DictReader(open(f[0])):
    if "march_kernel" in r["Name"] or "lattice_kernel" in r["Name"]:
        t[r["Name"][:62]] = (r["Calls"], float(r["AverageNs"]) / 1e3)
w = collections.defaultdict(dict)
for what in ("write", "fetch"):
    f = glob.glob("$OUT/$tag/%s/**/*counter_collection.csv" % what, recursive=True)
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:62]].append(float(r["Counter_Value"]))
    for k, vals in acc.items():
        w[k][what] = sum(vals[len(vals) // 2:]) / len(vals[len(vals) // 2:])
for k, (calls, us) in t.items():
    print("  %-62s calls %4s avg %7.2f us  write %8.0f KB fetch %8.0f KB" % (k, calls, us, w.get(k, {}).get("write", 0), w.get(k, {}).get("fetch", 0)))
PY
done 2>&1 | tee $OUT/summary.txt
