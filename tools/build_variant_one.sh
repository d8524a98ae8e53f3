#!/bin/bash
# Rebuild ONE source file with extra macros and link it with the regular objects of the others:
#   tools/build_variant_one.sh <file-stem> <name> "<flags>" [<name2> "<flags2>" ...]   -> build/variants/<name>.so
set -e
cd "$(dirname "$0")/../torchsparsegradutils_amd/csrc"
stem=$1; shift
mkdir -p ../../build/variants
others=$(ls *.o | grep -v "^$stem.o$")
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c $stem.hip -o ../../build/variants/${stem}_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others ../../build/variants/${stem}_$name.o -o ../../build/variants/$name.so && echo built $name ) &
done
wait
