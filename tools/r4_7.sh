#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_march.py -x -q -m gpu 2>&1 | tail -25 > $O/tests.log
for pat in per27 trunc27; do
  echo "== march $pat" >> $O/lb.log
  timeout 300 python tools/marchbench.py --pattern $pat --reps 30 --modes fwd,sddmm,spmmt,bwd >> $O/lb.log 2>&1
done
echo "== march per27 forced box arithmetic" >> $O/lb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck --force-rstart >> $O/lb.log 2>&1
echo "== fused backward configs per27" >> $O/lb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck --modes bwd --cfg 4,8,2,256 4,8,3,256 4,8,4,256 4,8,5,256 2,8,3,256 2,8,5,256 >> $O/lb.log 2>&1
grep -v amdgpu.ids $O/lb.log; tail -12 $O/tests.log
