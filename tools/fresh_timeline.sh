#!/bin/bash
# kernel timeline of the fresh-index-tensor steps (tools/fresh_profile.py): busy / idle GPU time per step and the kernels in one step
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp; rocprofv3 --kernel-trace --output-format csv -d /tmp/fp -o s -- python3 $GRAFT_REPO_ROOT/tools/fresh_profile.py > /tmp/fp.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/fp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the last 20 steps: find the last 20 forward march kernels
fw = [i for i, r in enumerate(rows) if "march_kernel<float, 8, 0" in r[2]]
i0 = fw[-20]
t_prev = rows[i0][0]
print("last steps: kernel, start offset us, duration us, gap before us")
for r in rows[fw[-3]:fw[-1]]:
    print("  %-60s %9.1f %8.1f" % (r[2][:60], (r[0] - rows[fw[-3]][0]) / 1e3, (r[1] - r[0]) / 1e3))
span = rows[fw[-1]][0] - rows[fw[-20]][0]
busy = sum(r[1] - r[0] for r in rows[fw[-20]:fw[-1]])
print("per step: span %.1f us, busy %.1f us" % (span / 19e3, busy / 19e3))
PY
