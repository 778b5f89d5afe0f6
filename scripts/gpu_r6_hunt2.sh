#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6hunt2; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python scripts/dev/rollout_race_hunt.py 3000 2>&1 | grep -v amdgpu.ids | cut -c1-300 > $O/rollout_hunt.txt; tail -5 $O/rollout_hunt.txt
timeout 1500 python scripts/dev/beh_race_hunt.py 12000 bwd 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/bwd_hunt.txt; tail -5 $O/bwd_hunt.txt
timeout 600 python scripts/dev/ppo_determinism.py > $O/ppo_determinism.txt 2>&1; tail -5 $O/ppo_determinism.txt
