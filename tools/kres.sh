#!/bin/bash
# Per-kernel register / LDS / occupancy table of one HIP source:  tools/kres.sh <file.hip> [extra hipcc flags]
cd "$(dirname "$0")/../torchsparsegradutils_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-pass-failed "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 |
  awk '/Function Name:/ {for(i=1;i<=NF;i++) if ($i ~ /^_Z/) name=$i}
       /remark: +(TotalSGPRs|VGPRs:|ScratchSize|Occupancy)/ {sub(/.*remark: +/,""); sub(/ \[-Rpass.*/,""); v=v" "$0";"}
       /LDS Size/ {cmd="echo "name" | c++filt"; cmd | getline d; close(cmd); sub(/tsgu::/,"",d); sub(/\(tsgu::.*/,"",d); print d" |"v; v=""}'
