#!/bin/bash
# A/B (same box): tail parameter vectors staged in LDS (both forward forms).  old = build/abl/lib_stagedw.so
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/abl; O=gpurun_out; export TMPDIR=/tmp
OLD=$R/build/abl/lib_stagedw.so
for i in 1 2; do
IPLAN_HIP_LIB=$OLD timeout 300 python scripts/microbench.py select_actions rollout ppo_train ac_train_parts ac_phases > $O/ab_old$i.log 2>&1
timeout 300 python scripts/microbench.py select_actions rollout ppo_train ac_train_parts ac_phases > $O/ab_new$i.log 2>&1
done
grep -H "gpu \|phases" $O/ab_old*.log $O/ab_new*.log > $O/abl_summary.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.log 2> $O/bench.err
IPLAN_HIP_LIB=$OLD IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_old.log 2> $O/bench_old.err
