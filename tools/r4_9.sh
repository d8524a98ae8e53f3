#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_march.py -x -q -m gpu 2>&1 | tail -8 > $O/tests.log
echo "== per27 uniform / forced box / trunc27" >> $O/lb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck >> $O/lb.log 2>&1
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck --force-rstart >> $O/lb.log 2>&1
timeout 300 python tools/marchbench.py --pattern trunc27 --reps 30 --nocheck >> $O/lb.log 2>&1
echo "== per27 sddmm / spmmt alternating configs" >> $O/lb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 40 --nocheck --modes sddmm,spmmt --cfg 8,8,3,512 4,8,3,256 8,8,3,512 4,8,3,256 8,8,3,512 4,8,3,256 4,8,4,256 4,8,5,256 4,8,4,256 4,8,5,256 8,8,6,512 8,8,6,512 >> $O/lb.log 2>&1
echo "== per27 fwd alternating configs" >> $O/lb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 40 --nocheck --modes fwd --cfg 4,8,3,256 4,8,5,256 4,8,3,256 4,8,5,256 4,8,6,256 4,8,3,256 4,8,6,256 8,8,3,512 8,8,3,512 >> $O/lb.log 2>&1
grep -v amdgpu.ids $O/lb.log; tail -5 $O/tests.log
