#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4h; mkdir -p $O
echo "== per27 uniform" >> $O/lb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck >> $O/lb.log 2>&1
echo "== per27 forced box" >> $O/lb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck --force-rstart >> $O/lb.log 2>&1
for v in 1 2 3; do
echo "== per27 forced box dbg$v" >> $O/lb.log
TSGU_LIB_PATH=$PWD/build/variants/dbg$v.so timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck --force-rstart >> $O/lb.log 2>&1
done
echo "== per27 nseg sweep" >> $O/lb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck --modes fwd --cfg 4,8,2,256 4,8,3,256 4,8,4,256 4,8,5,256 4,8,6,256 >> $O/lb.log 2>&1
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck --modes sddmm,spmmt --cfg 8,8,2,512 8,8,3,512 8,8,4,512 8,8,5,512 8,8,6,512 4,8,3,256 4,8,5,256 >> $O/lb.log 2>&1
grep -v amdgpu.ids $O/lb.log
