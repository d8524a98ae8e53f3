#!/bin/bash
# HBM traffic of the bench kernels: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — never combined with
# other trace domains) around `bench.py`, restricted to the tsgu kernels.  Run on the GPU box:
#   bash tools/prof_bench_pmc.sh ; python tools/pmc_summary.py gpurun_out/pmc_bench tsgu
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bench
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass
  timeout 600 rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "tsgu::csr_(spmm|blocktile|rowpack|mm_backward)" --output-format csv \
     -d $OUT/$1 -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$1.log 2>&1
done
find $OUT -name "*counter_collection.csv" | head
