#!/bin/bash
mkdir -p gpurun_out
{
echo "=== fwd-only probe"; timeout 300 python tools/fwd_only_probe.py 2>&1 | grep -v Warn | tail -6
echo "=== tilebench cold default"; timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "round 1" | grep tile
echo "=== tilebench cold nt perm"; TSGU_LIB_PATH=$PWD/build/variants/nt_words.so timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "check|round 1" | grep -E "check|tile"
echo "=== round5 tests"; timeout 1200 python -m pytest tests/test_gpu_round5.py -q 2>&1 | tail -4
} > gpurun_out/check_r5h.txt 2>&1
grep -v amdgpu.ids gpurun_out/check_r5h.txt | cut -c1-2500 | tail -40
