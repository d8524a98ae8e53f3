#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4m; mkdir -p $O
timeout 300 python tools/marchbench.py --pattern per27 --reps 40 --nocheck --ab-rows >> $O/lb.log 2>&1
timeout 300 python tools/marchbench.py --pattern trunc27 --reps 40 >> $O/lb.log 2>&1
timeout 900 python -m pytest tests/test_gpu_march.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -12 > $O/tests.log
grep -v amdgpu.ids $O/lb.log | grep -v "^rows of\|^pattern"; tail -6 $O/tests.log
