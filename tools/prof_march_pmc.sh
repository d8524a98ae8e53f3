#!/bin/bash
# PMC passes for the plane-march kernels of tools/marchbench.py (run on the GPU box):  tools/prof_march_pmc.sh <mode: fwd|sddmm|spmmt>
set -u
M=${1:-fwd}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_march_$M
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "march_kernel" --output-format csv -d $OUT/$name -o p -- python3 $ROOT/tools/marchbench.py --modes $M --reps 3 --nocheck > $OUT/$name.log 2>&1
}
run a GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
python3 $ROOT/tools/pmc_summary.py $OUT "march_kernel" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
