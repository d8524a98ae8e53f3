#!/bin/bash
# round 4, first GPU pass: the generalised plane march — tests, then timings per pattern
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_march.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r4a/march_tests.log
for pat in per27 trunc27 per7 trunc7 lower27 slower27 xper27; do
  timeout 300 python tools/marchbench.py --pattern $pat --reps 30 >> gpurun_out/r4a/marchbench.log 2>&1
done
timeout 300 python tools/marchbench.py --pattern trunc27 --rhs 128 --reps 20 --nocheck >> gpurun_out/r4a/marchbench.log 2>&1
timeout 300 python tools/marchbench.py --pattern per27 --rhs 64 --reps 20 --nocheck >> gpurun_out/r4a/marchbench.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
tail -c 1500 gpurun_out/r4a/march_tests.log; cat gpurun_out/r4a/marchbench.log | tail -60
