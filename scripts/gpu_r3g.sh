#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3g; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 800 python scripts/ppo_grad_error_table.py --top 4 --cases cfg3_1,cfg3_2,cfg3_3,n55_rows2250,n9_rows22950,switches,mb3x2 --json $O/ppo_grad_table.json > $O/ppo_grad_table.txt 2>&1
grep -v amdgpu $O/ppo_grad_table.txt | cut -c1-420
