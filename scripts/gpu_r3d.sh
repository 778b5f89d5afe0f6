#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3d; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python scripts/dev/ppo_bwd_probe.py > $O/probe_rows2250.txt 2>&1
timeout 600 python scripts/dev/ppo_bwd_probe.py cfg3 > $O/probe_cfg3.txt 2>&1
cat $O/probe_rows2250.txt $O/probe_cfg3.txt | grep -v amdgpu | cut -c1-250
