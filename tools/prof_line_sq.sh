#!/bin/bash
# Where the whole-line march kernels' wave cycles go (C5 steps, tools/pattern_steps.py c5): SQ / LDS counters in separate PMC passes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_line
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
            "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --kernel-include-regex "${KREGEX:-linemarch_}" --output-format csv -d $OUT/p$i -o p -- python3 $ROOT/tools/pattern_steps.py ${PATTERN:-c5} 4 > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python3 $ROOT/tools/pmc_summary.py $OUT ${KREGEX:-linemarch_} > $OUT/summary.txt
cat $OUT/summary.txt
