#!/bin/bash
# Round-5 GPU-box check #5: tile kernels after the LDS overflow fix (+ SDDMM A/B), PMC, tests, bench.
mkdir -p gpurun_out
{
echo "=== tilebench cold"; timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "check|round 1"
echo "=== tilebench cold, serial SDDMM reads"; TSGU_LIB_PATH=$PWD/build/variants/sddmm_serial.so timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "check|round 1" | grep -E "check|tile"
echo "=== tilebench warm"; timeout 600 python tools/tilebench.py 2>&1 | grep -E "round 1" | grep tile
echo "=== PMC"; bash tools/prof_tile_pmc.sh r5e 2>&1 | grep -A14 "tile_kernel" | head -50
echo "=== all gpu tests"; timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|FAILED" | tail -25
echo "=== bench"; timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r5e.json 2> gpurun_out/bench_r5e.err; tail -c 300 gpurun_out/bench_r5e.json
} > gpurun_out/check_r5e.txt 2>&1
grep -v amdgpu.ids gpurun_out/check_r5e.txt | cut -c1-3000 | tail -150
