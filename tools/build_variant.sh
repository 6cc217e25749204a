#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> [-Dflags...]  -> mv2d_amd/lib/variants/lib<name>.so (other objects reused)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p mv2d_amd/lib/variants
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value "$@" -c mv2d_amd/csrc/$src -o mv2d_amd/lib/variants/${name}_$base.o
objs=$(ls mv2d_amd/lib/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs mv2d_amd/lib/variants/${name}_$base.o -o mv2d_amd/lib/variants/lib$name.so
rm mv2d_amd/lib/variants/${name}_$base.o
echo built mv2d_amd/lib/variants/lib$name.so
