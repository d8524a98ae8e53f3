#!/bin/bash
# Build A/B variants of libtsgu_hip.so with different tuning macros into build/variants/<name>.so
# usage: tools/build_variants.sh name "-DTSGU_X=1 ..." [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/../torchsparsegradutils_amd/csrc"
mkdir -p ../../build/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  dir=../../build/variants/obj_$name; mkdir -p $dir
  for f in *.hip; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c $f -o $dir/${f%.hip}.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $dir/*.o -o ../../build/variants/$name.so
  echo built $name
done
