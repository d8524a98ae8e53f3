L="python tools/latbench.py --reps 20"
{
echo "=== product"; timeout 600 $L --cfg 8,8,3,512,4 8,8,6,512,4 8,8,3,512,5 16,8,6,1024,4
echo "=== nodma"; TSGU_LIB_PATH=build/variants/lat_nodma.so timeout 300 $L --nocheck --cfg 8,8,3,512,4
echo "=== tests"; timeout 900 python -m pytest tests/test_gpu_lattice.py -x -q 2>&1 | tail -15
} > gpurun_out/lat6.txt 2>&1
grep -v amdgpu.ids gpurun_out/lat6.txt | tail -40
bash tools/prof_lat_pmc.sh fwd 8,8,3,512,4 2>&1 | grep -E "INSTS|ACTIVE|WAIT|WAVE_CYC|LDS_IDX"
