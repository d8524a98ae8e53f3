#!/bin/bash
# Round-5 GPU-box check: full-size identity of the tile kernels with the plan-free kernels (+ timings), the whole GPU suite, smoke, bench.
mkdir -p gpurun_out
{
echo "=== tilebench cold (bit-identity checks first)"; timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "check|round 1"
echo "=== all gpu tests"; timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|FAILED" | tail -25
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "=== bench"; SECONDS=0; timeout 1200 python bench.py > gpurun_out/bench_r5.json 2> gpurun_out/bench_r5.err; echo "bench wall seconds: $SECONDS"; tail -c 300 gpurun_out/bench_r5.json
} > gpurun_out/check_r5.txt 2>&1
grep -v amdgpu.ids gpurun_out/check_r5.txt | cut -c1-3000 | tail -60
