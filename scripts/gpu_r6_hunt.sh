#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6hunt; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python scripts/dev/beh_race_hunt.py ${REPS:-1500} ${STAGE:-both} 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/hunt.txt
tail -12 $O/hunt.txt
timeout 300 python scripts/microbench.py behavior_learn rollout 2>&1 | grep -v amdgpu.ids | tee $O/mb.txt
