#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for M in spmmt fwd; do
OUT=$ROOT/gpurun_out/r4l/pmc_$M
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "march_kernel" --output-format csv -d $OUT/$name -o p -- python3 $ROOT/tools/marchbench.py --modes $M --reps 3 --nocheck --ab-rows > $OUT/$name.log 2>&1
}
run a GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
python3 $ROOT/tools/pmc_summary.py $OUT "march_kernel" > $OUT/summary.txt 2>&1
echo "=== $M"; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +200k -delete
done
