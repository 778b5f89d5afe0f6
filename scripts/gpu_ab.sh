#!/bin/bash
# A/B of kernel variants: GPU parity tests on the default build, then microbench on both builds.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python scripts/microbench.py $MB_PIECES > gpurun_out/mb_default.log 2>&1
if [ -n "$ALT_LIB" ]; then IPLAN_HIP_LIB=$R/$ALT_LIB timeout 600 python scripts/microbench.py $MB_PIECES > gpurun_out/mb_alt.log 2>&1; fi
