{
echo "=== all gpu tests"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== bench"; timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1
} > gpurun_out/lat10.txt 2>&1
grep -v amdgpu.ids gpurun_out/lat10.txt | cut -c1-6000 | tail -30
