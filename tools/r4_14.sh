#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4n; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|took|Error|FAILED" $O/tests.log | head -30
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4n/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['frac_of_hbm_peak'], d['kernels_ms_in_step'], d['roofline']['kernel'][:40], d['roofline']['frac'], d['roofline']['frac_wire'])
for k,v in d['patterns'].items(): print(k, {a:v.get(a) for a in ('kernels','ms_per_step','frac','error')})
print(d['c5']['fwd_compute_only'], d['c5']['fwd_bwd_compute_only'])
PY
