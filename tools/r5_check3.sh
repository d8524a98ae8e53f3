#!/bin/bash
# Round-5 GPU-box check #3: tile schedule A/B + PMC, graph probe, tests, C3/C4 configs, bench.
mkdir -p gpurun_out
{
echo "=== round5 tests"; timeout 1200 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -8
echo "=== tilebench cold, cyclic schedule"; TSGU_TILE_CYCLIC=1 timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "check|round 1" | grep -E "check|tile"
echo "=== tilebench cold, run-per-workgroup schedule"; TSGU_TILE_CYCLIC=0 timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "round 1" | grep tile
echo "=== tilebench warm, cyclic"; TSGU_TILE_CYCLIC=1 timeout 600 python tools/tilebench.py 2>&1 | grep -E "round 1" | grep tile
echo "=== PMC cyclic"; TSGU_TILE_CYCLIC=1 bash tools/prof_tile_pmc.sh cyclic 2>&1 | tail -60
echo "=== PMC chunked"; TSGU_TILE_CYCLIC=0 bash tools/prof_tile_pmc.sh chunked 2>&1 | grep -A12 "tile_kernel" | head -60
echo "=== graph probe"; timeout 600 python tools/graph_probe.py 2>&1 | grep -v Warn | tail -6
echo "=== all gpu tests"; timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|FAILED" | tail -25
echo "=== configs C3 C4"; timeout 900 python bench_configs.py --only c3,c4 --no-cpu 2>&1 | grep -v Warn | tail -4
echo "=== bench"; timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r5c.json 2> gpurun_out/bench_r5c.err; tail -c 300 gpurun_out/bench_r5c.json
} > gpurun_out/check_r5c.txt 2>&1
grep -v amdgpu.ids gpurun_out/check_r5c.txt | cut -c1-3000 | tail -150
