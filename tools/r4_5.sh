#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4e; mkdir -p $O
python -m pytest tests/test_gpu_march.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -25 > $O/tests.log
for pat in per27 trunc27 per7 trunc7 lower27; do
  echo "== sweep $pat" >> $O/lb.log
  timeout 300 python tools/latbench.py --pattern $pat --reps 30 --nocheck >> $O/lb.log 2>&1
  echo "== march $pat" >> $O/lb.log
  timeout 300 python tools/marchbench.py --pattern $pat --reps 30 --nocheck >> $O/lb.log 2>&1
done
echo "== HEAD tree per27" >> $O/lb.log
(cd build/head_tree && timeout 300 python tools/marchbench.py --reps 30 --nocheck) >> $O/lb.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
grep -v amdgpu.ids $O/lb.log; tail -12 $O/tests.log; tail -3 $O/bench.err
