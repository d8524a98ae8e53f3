#!/bin/bash
# Round-5 GPU-box check #2: tile kernels (tests + timings), whole suite, bench.
mkdir -p gpurun_out
{
echo "=== round5 tests"; timeout 1200 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -15
echo "=== tilebench cold"; timeout 600 python tools/tilebench.py --cold 2>&1 | grep -v Warn | tail -30
echo "=== tilebench warm"; timeout 600 python tools/tilebench.py 2>&1 | grep -v Warn | grep "round 1" | tail -12
echo "=== host profile of the forward-only sparse_mm on the published rand shape"; timeout 300 python tools/fwd_only_profile.py 2>&1 | grep -v Warn | tail -25
echo "=== all gpu tests"; timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|FAILED" | tail -25
echo "=== bench"; timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r5b.json 2> gpurun_out/bench_r5b.err; tail -c 300 gpurun_out/bench_r5b.json
} > gpurun_out/check_r5b.txt 2>&1
grep -v amdgpu.ids gpurun_out/check_r5b.txt | cut -c1-3000 | tail -120
