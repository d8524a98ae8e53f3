#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4p; mkdir -p $O
for i in 1 2; do
echo "== product build" >> $O/lb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 40 --nocheck --modes fwd,spmmt >> $O/lb.log 2>&1
echo "== every wave takes the plain copy (wrong results on wrapped rows: upper bound of what a faster gather path can buy)" >> $O/lb.log
TSGU_LIB_PATH=$PWD/build/variants/allplain.so timeout 300 python tools/marchbench.py --pattern per27 --reps 40 --nocheck --modes fwd,spmmt >> $O/lb.log 2>&1
done
grep -v amdgpu.ids $O/lb.log | grep -v "^rows of\|^pattern"
