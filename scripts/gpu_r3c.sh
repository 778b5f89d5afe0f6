#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3c; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python scripts/ppo_grad_error_table.py --top 10 --cases n55_rows60,n55_rows2250,n9_rows22950 --json $O/ppo_grad_table_sizes.json > $O/ppo_grad_table_sizes.txt 2>&1
cat $O/ppo_grad_table_sizes.txt | cut -c1-160
