#!/bin/bash
# In-step durations of the three march kernels of a pattern with ONE product's configuration pinned (the others unchanged):
#   bash tools/march_instep_probe.sh <pattern> <FWD|SDDMM|SPMMT> "ty,tz,nseg,threads" ...     (rocprofv3 --stats around tools/pattern_steps.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PAT=$1; MODE=$2; shift 2
cd /tmp; export TMPDIR=/tmp
for cfg in default "$@"; do
  unset TSGU_MARCH_CFG_FWD TSGU_MARCH_CFG_SDDMM TSGU_MARCH_CFG_SPMMT
  [ "$cfg" != default ] && export TSGU_MARCH_CFG_$MODE=$cfg
  rm -rf /tmp/mip; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mip -o s -- python3 $ROOT/tools/pattern_steps.py $PAT 200 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
t = [(r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(glob.glob("/tmp/mip/**/*kernel_stats.csv", recursive=True)[0])) if "tsgu::" in r["Name"] and int(r["Calls"]) >= 200]
role = lambda n: {"0": "fwd", "1": "sddmm", "2": "spmmt"}.get(n.split("<", 1)[1].split(",")[2].strip(), "?") if "march_kernel" in n else n[12:40]
print("$MODE=$cfg", " ".join(f"{role(n)} {u:.1f}" for n, c, u in sorted(t, key=lambda x: role(x[0]))), "sum %.1f" % sum(u for n, c, u in t))
PY
done
