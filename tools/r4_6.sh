#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4f; mkdir -p $O
python -m pytest tests/test_gpu_march.py tests/test_gpu_round4.py tests/test_gpu_lattice.py -x -q -m gpu 2>&1 | tail -25 > $O/tests.log
for pat in per27 trunc27 xper27; do
  echo "== march $pat" >> $O/lb.log
  timeout 300 python tools/marchbench.py --pattern $pat --reps 30 --nocheck >> $O/lb.log 2>&1
done
echo "== march lower27 sddmm" >> $O/lb.log
timeout 300 python tools/marchbench.py --pattern lower27 --reps 30 --modes sddmm >> $O/lb.log 2>&1
echo "== HEAD tree per27" >> $O/lb.log
(cd build/head_tree && timeout 300 python tools/marchbench.py --reps 30 --nocheck) >> $O/lb.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -v amdgpu.ids $O/lb.log; tail -12 $O/tests.log; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['frac_of_hbm_peak'])
for k,v in d['patterns'].items(): print(k, {a:v.get(a) for a in ('kernels','ms_per_step','frac','error')})
PY
