#!/bin/bash
# PMC passes for one kernel of tools/kbench.py (run on the GPU box).
# usage: tools/prof_pmc.sh <kbench --only name> <kernel regex> [extra kbench args]
# Each pass is its own rocprofv3 run with --kernel-trace only; counters are collected only for kernels
# matching the regex so that the (slow under PMC) setup kernels run at full speed.
set -u
K=${1:-spmm}; RX=${2:-csr_spmm}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$K
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
EXTRA=("$@")
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$RX" --output-format csv -d $OUT/$name -o p -- python3 $ROOT/tools/kbench.py --only $K --reps 3 "${EXTRA[@]}" > $OUT/$name.log 2>&1
}
run fetch FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
python3 $ROOT/tools/pmc_summary.py $OUT "$RX" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
