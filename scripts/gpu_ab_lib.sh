#!/bin/bash
# A/B of build/abl/lib_*.so variants on microbench pieces: MB_PIECES="gat_fwd rollout" REPS=2 bash scripts/gpu_ab_lib.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/ab; mkdir -p $O; export TMPDIR=/tmp
: > $O/mb.txt
for rep in $(seq 1 ${REPS:-2}); do
for lib in $(ls build/abl/lib_*.so | sort -V); do
  echo "== $(basename $lib .so)" >> $O/mb.txt
  IPLAN_HIP_LIB=$R/$lib timeout 200 python scripts/microbench.py ${MB_PIECES:-gat_fwd} 2>&1 | grep -v amdgpu.ids >> $O/mb.txt
done; done
cat $O/mb.txt
