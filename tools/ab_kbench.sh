#!/bin/bash
# kbench with the default library and with every build/variants/*.so (GPU box):  tools/ab_kbench.sh <kbench args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
echo "=== default"; python3 $ROOT/tools/kbench.py "$@" 2>&1 | grep -E "GB/s|check"
for so in $ROOT/build/variants/*.so; do
  echo "=== $(basename $so .so)"; TSGU_LIB_PATH=$so python3 $ROOT/tools/kbench.py "$@" 2>&1 | grep -E "GB/s algorithmic|check"
done
