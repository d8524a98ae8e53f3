#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4c; mkdir -p $O
python -m pytest tests/test_gpu_march.py -x -q -m gpu 2>&1 | tail -15 > $O/march_tests.log
echo "== HEAD tree per27" >> $O/mb.log
(cd build/head_tree && timeout 300 python tools/marchbench.py --reps 30 --nocheck) >> $O/mb.log 2>&1
echo "== new per27 (two-phase gather)" >> $O/mb.log
timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck >> $O/mb.log 2>&1
echo "== new per27 (single-phase gather)" >> $O/mb.log
TSGU_LIB_PATH=$PWD/build/variants/g1.so timeout 300 python tools/marchbench.py --pattern per27 --reps 30 --nocheck >> $O/mb.log 2>&1
echo "== new trunc27 (two-phase / single-phase)" >> $O/mb.log
timeout 300 python tools/marchbench.py --pattern trunc27 --reps 30 >> $O/mb.log 2>&1
TSGU_LIB_PATH=$PWD/build/variants/g1.so timeout 300 python tools/marchbench.py --pattern trunc27 --reps 30 --nocheck >> $O/mb.log 2>&1
for pat in per7 trunc7 lower27 slower27 upper27 lower7; do
  echo "== new $pat" >> $O/mb.log
  timeout 300 python tools/marchbench.py --pattern $pat --reps 30 >> $O/mb.log 2>&1
done
echo "== per7 single-phase" >> $O/mb.log
TSGU_LIB_PATH=$PWD/build/variants/g1.so timeout 300 python tools/marchbench.py --pattern per7 --reps 30 --nocheck >> $O/mb.log 2>&1
echo "== xper27 (run-time... no: full box, mixed periodicity)" >> $O/mb.log
timeout 300 python tools/marchbench.py --pattern xper27 --reps 30 >> $O/mb.log 2>&1
echo "== HEAD tree per27 again" >> $O/mb.log
(cd build/head_tree && timeout 300 python tools/marchbench.py --reps 30 --nocheck) >> $O/mb.log 2>&1
grep -v amdgpu.ids $O/mb.log; tail -5 $O/march_tests.log
