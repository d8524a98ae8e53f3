#!/bin/bash
# Round-5 GPU-box check #7: full GPU suite, smoke, bench (timed), tilebench.
mkdir -p gpurun_out
{
echo "=== tilebench cold"; timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "check|round 1" | grep -E "check|tile"
echo "=== all gpu tests"; timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|FAILED" | tail -25
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "=== bench"; /usr/bin/time -v timeout 1200 python bench.py > gpurun_out/bench_r5g.json 2> gpurun_out/bench_r5g.err; grep -E "Elapsed|Maximum resident" gpurun_out/bench_r5g.err; tail -c 300 gpurun_out/bench_r5g.json
} > gpurun_out/check_r5g.txt 2>&1
grep -v amdgpu.ids gpurun_out/check_r5g.txt | cut -c1-3000 | tail -60
