#!/bin/bash
# Rebuild SEVERAL source files with extra macros and link them with the regular objects of the others:
#   tools/build_variant_multi.sh <name> "<flags>" <stem> [<stem> ...]   -> build/variants/<name>.so
set -e
cd "$(dirname "$0")/../torchsparsegradutils_amd/csrc"
name=$1; flags=$2; shift 2
mkdir -p ../../build/variants
others=$(ls *.o)
objs=""
for stem in "$@"; do
  others=$(echo "$others" | grep -v "^$stem.o$")
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c $stem.hip -o ../../build/variants/${stem}_$name.o &
  objs="$objs ../../build/variants/${stem}_$name.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $objs -o ../../build/variants/$name.so && echo built $name
