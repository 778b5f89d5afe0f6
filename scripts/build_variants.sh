#!/bin/bash
# Kernel A/B helper: scripts/build_variants.sh <file.hip> <MACRO> v1 v2 ...  ->  build/abl/lib_<v>.so
# (the named source compiled with -D<MACRO>=<v>, linked with the regular objects of the other sources)
set -e
cd "$(dirname "$0")/.."
src=$1; macro=$2; shift 2
make -s -C iplan_amd/csrc all
rm -rf build/abl; mkdir -p build/abl
objs=$(ls build/obj/*.hip.o build/obj/*.cpp.o | grep -v "/$src.o")
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iiplan_amd/csrc -Wno-unused-result -D$macro=$v -x hip -c iplan_amd/csrc/$src -o build/abl/v_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/abl/v_$v.o -o build/abl/lib_$v.so ) &
done
wait
rm -f build/abl/*.o; ls build/abl
