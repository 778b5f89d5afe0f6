#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3e; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 800 python scripts/dev/ppo_row_probe.py > $O/row_probe_cfg3_nosync.txt 2>&1
grep -v amdgpu $O/row_probe_cfg3_nosync.txt | grep -h "grad \|last_step\|oracle\|indep\|old_logp\|Error\|error" | cut -c1-220
