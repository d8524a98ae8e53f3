#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one kbench kernel (two separate PMC passes).  usage: prof_kbench_traffic.sh <only> <regex> [kbench args]
set -u
K=$1; RX=$2; shift 2; EXTRA=("$@")
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic_$K
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "write WRITE_SIZE GRBM_GUI_ACTIVE"; do
  set -- $pass
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$RX" --output-format csv -d $OUT/$name -o p -- python3 $ROOT/tools/kbench.py --only $K --reps 3 --rpb "${EXTRA[@]}" > $OUT/$name.log 2>&1
done
