#!/bin/bash
# L2 / write-path PMC passes for one kernel of tools/kbench.py (run on the GPU box).
# usage: [TSGU_LIB_PATH=...] tools/prof_pmc_l2.sh <tag> <kbench --only name> <kernel regex> [extra kbench args]
set -u
TAG=$1; K=$2; RX=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcl2_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
EXTRA=("$@")
run() {
  local name=$1; shift
  timeout 90 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$RX" --output-format csv -d $OUT/$name -o p -- python3 $ROOT/tools/kbench.py --only $K --reps 3 "${EXTRA[@]}" > $OUT/$name.log 2>&1
}
# at most four TCC counters fit one pass (more: "exceeds the capabilities of the hardware", and the aborted run hangs
# until the timeout) — keep the passes small and the timeout short
run l2w TCC_REQ_sum TCC_WRITE_sum TCC_TAG_STALL_sum TCP_TCC_WRITE_REQ_sum
run l2e TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCP_TCC_WRITE_REQ_LATENCY_sum
run l2r TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
python3 $ROOT/tools/pmc_summary.py $OUT "$RX" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
