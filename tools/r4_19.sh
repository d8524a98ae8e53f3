#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4s; mkdir -p $O
timeout 300 python tools/host_cprofile.py 2000 > $O/host.log 2>&1
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "capturable" 2>&1 | tail -5 >> $O/host.log
grep -v "amdgpu.ids" $O/host.log | cut -c1-180 | head -70
