#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for i in 1 2; do
IPLAN_NO_FC1_PACK=1 timeout 200 python scripts/microbench.py select_actions ac_phases rollout ppo_train ac_train_parts > $O/ab_nopack$i.log 2>&1
timeout 200 python scripts/microbench.py select_actions ac_phases rollout ppo_train ac_train_parts > $O/ab_pack$i.log 2>&1
done
IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.log 2> $O/bench.err
IPLAN_NO_FC1_PACK=1 IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_nopack.log 2> $O/bench_nopack.err
