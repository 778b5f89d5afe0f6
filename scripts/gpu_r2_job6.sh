#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for i in 1 2; do
IPLAN_HIP_LIB=$R/build/abl/lib_nostage.so timeout 200 python scripts/microbench.py select_actions ac_phases rollout > $O/ab_nostage$i.log 2>&1
timeout 200 python scripts/microbench.py select_actions ac_phases rollout > $O/ab_stage$i.log 2>&1
done
