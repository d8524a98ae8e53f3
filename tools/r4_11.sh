#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4k; mkdir -p $O
timeout 300 python tools/marchbench.py --pattern per27 --reps 40 --nocheck --ab-rows >> $O/lb.log 2>&1
timeout 300 python tools/marchbench.py --pattern per27 --reps 40 --nocheck --modes sddmm --cfg 4,8,3,256 4,8,4,256 4,8,3,256 4,8,4,256 8,8,3,512 4,8,3,256 >> $O/lb.log 2>&1
grep -v amdgpu.ids $O/lb.log | grep -v "^rows of\|^pattern"
