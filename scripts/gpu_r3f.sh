#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3f; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 800 python scripts/dev/ppo_post_probe.py > $O/post_probe.txt 2>&1
grep -v amdgpu $O/post_probe.txt | cut -c1-220 | tail -70
