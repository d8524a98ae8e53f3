#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4d; mkdir -p $O
python -m pytest tests/test_gpu_march.py -x -q -m gpu 2>&1 | tail -15 > $O/march_tests.log
for pat in per7 trunc7 lower27 trunc27; do
  echo "== sweep $pat" >> $O/lb.log
  timeout 300 python tools/latbench.py --pattern $pat --reps 30 --nocheck >> $O/lb.log 2>&1
  echo "== march $pat" >> $O/lb.log
  timeout 300 python tools/marchbench.py --pattern $pat --reps 30 --nocheck >> $O/lb.log 2>&1
done
grep -v amdgpu.ids $O/lb.log; tail -5 $O/march_tests.log
