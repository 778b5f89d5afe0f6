#!/bin/bash
# Whole-library A/B: scripts/build_ab.sh NAME "-DFLAG ..."  ->  build/abl/lib_NAME.so (all sources rebuilt with the flags)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
make -s -C iplan_amd/csrc all OBJDIR=$PWD/build/ab_$1 OUT=$PWD/build/abl/lib_$1.so EXTRA="$2"
rm -rf build/ab_$1
ls -la build/abl/lib_$1.so
