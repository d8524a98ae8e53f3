#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4r; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/tests.log
bash tools/prof_round.sh r04 > $O/prof.log 2>&1
# keep the merged output small: drop the big traces of the pattern passes (their counter CSVs are what is read)
find gpurun_out/prof_r04 -name "*kernel_trace.csv" -path "*pat_*" -delete
find gpurun_out/prof_r04 -name "*.csv" -size +8M -delete
du -sh gpurun_out/prof_r04
grep -E "passed|failed|FAILED" $O/tests.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/prof_r04/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['frac_of_hbm_peak'], d['kernels_ms_in_step'], d['host_ms_per_step'], d['cpu_baseline']['value'])
for k,v in d['patterns'].items(): print(k, {a:v.get(a) for a in ('kernels','ms_per_step','frac','error')})
PY
