#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3h; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dp_gpu_cycle.py -m gpu -q -x > $O/pytest_dp_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dp_gpu.log
grep -n "Error\|rank.\]:" $O/pytest_dp_gpu.log | head -30 | cut -c1-250; tail -3 $O/pytest_dp_gpu.log
