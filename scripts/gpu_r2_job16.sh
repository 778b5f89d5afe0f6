#!/bin/bash
# gat_bwd: step-ahead record fetch + batched gather (old = build/abl/lib_gatbwd_old.so)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out; export TMPDIR=/tmp
OLD=$R/build/abl/lib_gatbwd_old.so
P="gat_bwd_phases gat_fwd(save)+bwd+wgrad_S=64 prediction_learn"
for i in 1 2; do
IPLAN_HIP_LIB=$OLD timeout 200 python scripts/microbench.py gat_bwd_phases "gat_fwd(save)+bwd+wgrad S=64" "gat_fwd(save) S=64" prediction_learn behavior_learn > $O/ab_old$i.log 2>&1
timeout 200 python scripts/microbench.py gat_bwd_phases "gat_fwd(save)+bwd+wgrad S=64" "gat_fwd(save) S=64" prediction_learn behavior_learn > $O/ab_new$i.log 2>&1
done
grep -H "gpu \|phases" $O/ab_old*.log $O/ab_new*.log > $O/abl_summary.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python scripts/cfg5_bench.py --pieces gat >> $O/abl_summary.txt 2>&1
IPLAN_HIP_LIB=$OLD timeout 300 python scripts/cfg5_bench.py --pieces gat >> $O/abl_summary.txt 2>&1
