#!/bin/bash
# PMC passes for the lattice kernels of tools/latbench.py (run on the GPU box).
# usage: tools/prof_lat_pmc.sh <mode: fwd|sddmm|spmmt> <cfg ty,tz,nseg,threads,ring> [extra latbench args]
set -u
M=${1:-fwd}; CFG=${2:-8,8,3,512,4}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lat_$M
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
EXTRA=("$@")
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "lattice_kernel" --output-format csv -d $OUT/$name -o p -- python3 $ROOT/tools/latbench.py --modes $M --cfg $CFG --reps 3 --nocheck "${EXTRA[@]}" > $OUT/$name.log 2>&1
}
run fetch FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
python3 $ROOT/tools/pmc_summary.py $OUT "lattice_kernel" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
