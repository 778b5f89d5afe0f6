#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; export TMPDIR=/tmp
bash scripts/gpu_r6_ab_beh.sh > /dev/null 2>&1
paste - - - < gpurun_out/r6ab/ab.txt | cut -c1-200
REPS=24000 STAGE=fwd bash scripts/gpu_r6_hunt.sh
