#!/bin/bash
# One GPU-box round of checks: lattice tests, the whole GPU suite, smoke, a short bench line.  Output in gpurun_out/check.txt
{
echo "=== lattice + march tests"; timeout 900 python -m pytest tests/test_gpu_lattice.py tests/test_gpu_march.py -x -q 2>&1 | tail -3
echo "=== all gpu tests"; timeout 3400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -6
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== bench"; timeout 900 python bench.py --steps 20 --warmup 5 "$@" 2>&1 | tail -1
} > gpurun_out/check.txt 2>&1
grep -v amdgpu.ids gpurun_out/check.txt | cut -c1-2500 | tail -30
