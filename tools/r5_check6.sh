#!/bin/bash
# Round-5 GPU-box check #6: tile kernels with aligned entry-byte reads, tests, bench.
mkdir -p gpurun_out
{
echo "=== tilebench cold"; timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "check|round 1"
echo "=== tilebench warm"; timeout 600 python tools/tilebench.py 2>&1 | grep -E "round 1" | grep tile
echo "=== PMC"; bash tools/prof_tile_pmc.sh r5f 2>&1 | grep -A14 "tile_kernel" | head -50
echo "=== round5 + march + lattice tests"; timeout 3000 python -m pytest tests/test_gpu_round5.py tests/test_gpu_march.py tests/test_gpu_round4.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|FAILED" | tail -25
echo "=== bench"; timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r5f.json 2> gpurun_out/bench_r5f.err; tail -c 300 gpurun_out/bench_r5f.json
} > gpurun_out/check_r5f.txt 2>&1
grep -v amdgpu.ids gpurun_out/check_r5f.txt | cut -c1-3000 | tail -150
