#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4b; mkdir -p $O
python -m pytest tests/test_gpu_march.py -x -q -m gpu 2>&1 | tail -15 > $O/march_tests.log
echo "== HEAD tree per27" >> $O/mb.log
(cd build/head_tree && timeout 300 python tools/marchbench.py --reps 30 --nocheck) >> $O/mb.log 2>&1
echo "== new per27" >> $O/mb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck >> $O/mb.log 2>&1
echo "== new per27 forced row pointers" >> $O/mb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --force-rstart >> $O/mb.log 2>&1
for pat in trunc27 per7 trunc7 lower27; do
  echo "== new $pat" >> $O/mb.log
  timeout 300 python tools/marchbench.py --pattern $pat --reps 30 >> $O/mb.log 2>&1
done
echo "== trunc27 256-thread configs" >> $O/mb.log
timeout 300 python tools/marchbench.py --pattern trunc27 --reps 30 --nocheck --cfg 4,8,3,256 4,8,4,256 8,8,4,512 >> $O/mb.log 2>&1
echo "== per7 configs" >> $O/mb.log
timeout 300 python tools/marchbench.py --pattern per7 --reps 30 --nocheck --cfg 4,8,3,256 4,8,5,256 8,8,3,512 8,8,5,512 >> $O/mb.log 2>&1
echo "== HEAD tree per27 again" >> $O/mb.log
(cd build/head_tree && timeout 300 python tools/marchbench.py --reps 30 --nocheck) >> $O/mb.log 2>&1
grep -v amdgpu.ids $O/mb.log; tail -5 $O/march_tests.log
