#!/bin/bash
# Everything profiles/ needs for one round, on the GPU box:   bash tools/prof_round.sh <tag>     (e.g. r02)
#   1. rocprofv3 --kernel-trace --stats of the default bench command            -> gpurun_out/prof_<tag>/stats
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) -> gpurun_out/prof_<tag>/{fetch,write}
#   3. the un-profiled bench line and the secondary configurations               -> gpurun_out/prof_<tag>/*.json(l)
# Copy the summaries into profiles/ with tools/collect_profiles.py afterwards.
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-c5 > $OUT/stats.log 2>&1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass
  timeout 900 rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "tsgu::(march_kernel|lattice_kernel|tile_kernel|csr_(spmm|sddmm|rowpack|mm_backward))" --output-format csv \
     -d $OUT/$1 -o p -- python3 $ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-c5 --no-patterns > $OUT/$1.log 2>&1
done
# per-pattern HBM bytes per step (bench.py's `patterns` block): the same two passes around a few steps of each pattern
for pat in headline c2_7pt_periodic c2_27pt_truncated c2_27pt_truncated_lower mesh27_blocked cfd2_shaped cfd2_mesh c5; do
  # (durations of the pattern's kernels: their own --stats pass, 100 steps — counters and timing never share a run)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pat_$pat/stats -o s -- python3 $ROOT/tools/pattern_steps.py $pat 100 > $OUT/pat_$pat.stats.log 2>&1
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
    set -- $pass
    timeout 600 rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "tsgu::" --output-format csv \
       -d $OUT/pat_$pat/$1 -o p -- python3 $ROOT/tools/pattern_steps.py $pat 6 > $OUT/pat_$pat.$1.log 2>&1
  done
done
timeout 900 python3 $ROOT/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
find $OUT -name "*.csv" | head -20
