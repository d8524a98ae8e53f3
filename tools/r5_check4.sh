#!/bin/bash
# Round-5 GPU-box check #4: trimmed tile kernels + PMC, cfd2 gather-depth variants, graph probe, fresh-tensor host profile, tests, bench.
mkdir -p gpurun_out
{
echo "=== tilebench cold"; timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "check|round 1"
echo "=== tilebench warm"; timeout 600 python tools/tilebench.py 2>&1 | grep -E "round 1" | grep tile
echo "=== PMC"; bash tools/prof_tile_pmc.sh r5d 2>&1 | grep -A14 "tile_kernel" | head -70
echo "=== cfd2 variants"; timeout 300 python tools/cfd2bench.py 2>&1 | grep -v Warn | tail -4
for so in build/variants/*.so; do TSGU_LIB_PATH=$PWD/$so timeout 300 python tools/cfd2bench.py 2>&1 | grep -v Warn | tail -4; done
echo "=== graph probe"; timeout 600 python tools/graph_probe.py 2>&1 | grep -v Warn | tail -6
echo "=== fresh tensors host profile"; timeout 300 python tools/fresh_profile.py 2>&1 | grep -v Warn | tail -32
echo "=== all gpu tests"; timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|FAILED" | tail -25
echo "=== bench"; timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r5d.json 2> gpurun_out/bench_r5d.err; tail -c 300 gpurun_out/bench_r5d.json
} > gpurun_out/check_r5d.txt 2>&1
grep -v amdgpu.ids gpurun_out/check_r5d.txt | cut -c1-3000 | tail -150
