#!/bin/bash
# block order experiment: tile kernels with the natural and the cube-by-cube block order
mkdir -p gpurun_out
{
for o in 0 1; do echo "=== TSGU_TILE_ORDER=$o"; TSGU_TILE_ORDER=$o timeout 600 python tools/tilebench.py --cold 2>&1 | grep -E "check|round|tile plan" | head -30; done
} > gpurun_out/order_probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/order_probe.txt | cut -c1-600
bash tools/r5_check.sh
