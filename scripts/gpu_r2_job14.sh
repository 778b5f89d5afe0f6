#!/bin/bash
# A/B (same box): XCD-aware workgroup placement in the actor/critic forward.  old = build/abl/lib_noremap.so
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/abl; O=gpurun_out; export TMPDIR=/tmp
OLD=$R/build/abl/lib_noremap.so
for i in 1 2 3; do
IPLAN_HIP_LIB=$OLD timeout 300 python scripts/microbench.py select_actions rollout ppo_train ac_train_parts ac_phases > $O/ab_old$i.log 2>&1
timeout 300 python scripts/microbench.py select_actions rollout ppo_train ac_train_parts ac_phases > $O/ab_new$i.log 2>&1
done
grep -H "gpu \|phases" $O/ab_old*.log $O/ab_new*.log | grep -v "infer\|save)" > $O/abl_summary.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
