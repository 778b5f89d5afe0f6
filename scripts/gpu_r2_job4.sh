#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
IPLAN_HIP_LIB=$R/build/abl/lib_gatclk.so timeout 200 python scripts/microbench.py gat_fwd gat_phases12 > $O/ab_gatclk.log 2>&1
for v in pf2 pf5; do IPLAN_HIP_LIB=$R/build/abl/lib_$v.so timeout 200 python scripts/microbench.py select_actions ac_phases rollout > $O/ab_$v.log 2>&1; done
timeout 200 python scripts/microbench.py select_actions ac_phases rollout > $O/ab_base.log 2>&1
