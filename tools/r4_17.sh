#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_march.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -8 > $O/tests.log
for i in 1 2; do
timeout 300 python tools/marchbench.py --pattern per27 --reps 40 >> $O/lb.log 2>&1
timeout 300 python tools/marchbench.py --pattern trunc27 --reps 40 >> $O/lb.log 2>&1
done
echo "== HEAD tree per27" >> $O/lb.log
(cd build/head_tree && timeout 300 python tools/marchbench.py --reps 40 --nocheck) >> $O/lb.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c5 > $O/bench.json 2> $O/bench.err
grep -v amdgpu.ids $O/lb.log | grep -v "^rows of\|^pattern\|^plan"; tail -4 $O/tests.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4q/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['frac_of_hbm_peak'], d['kernels_ms_in_step'], d['roofline']['kernel'][:40], d['roofline']['frac'])
for k,v in d['patterns'].items(): print(k, {a:v.get(a) for a in ('kernels','ms_per_step','frac','error')})
PY
