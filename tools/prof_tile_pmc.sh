#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit counters of the tile kernels (separate PMC passes).  usage: prof_tile_pmc.sh <tag> [tilebench args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_tile_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE GRBM_GUI_ACTIVE" "write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT"; do
  set -- $pass
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "tile_kernel|csr_rowpack_kernel" --output-format csv -d $OUT/$name -o p -- python3 $ROOT/tools/tilebench.py --reps 3 > $OUT/$name.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT
