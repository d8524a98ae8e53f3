#!/bin/bash
# Round-5 GPU-box check: new tests first, tolerance probe, the whole GPU suite, smoke, bench line.  Output under gpurun_out/
mkdir -p gpurun_out
{
echo "=== round5 tests"; timeout 1200 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -15
echo "=== tri tolerance probe"; timeout 600 python tools/tri_tol_probe.py 2>&1 | grep -v Warn | tail -40
echo "=== all gpu tests"; timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|FAILED" | tail -25
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "=== bench"; timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r5a.json 2> gpurun_out/bench_r5a.err; tail -c 600 gpurun_out/bench_r5a.json
} > gpurun_out/check_r5a.txt 2>&1
grep -v amdgpu.ids gpurun_out/check_r5a.txt | cut -c1-3000 | tail -80
