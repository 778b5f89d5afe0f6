#!/bin/bash
# deferred decoder update (wgrad + clip + Adam beside the next rollout): tests + same-box A/B of the whole cycle
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $O/pytest_gpu.log
for i in 1 2; do
IPLAN_NO_DEFER_DECODER=1 IPLAN_BENCH_WATCHDOG=120 timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_inline$i.log 2> $O/bench_inline$i.err < /dev/null
IPLAN_BENCH_WATCHDOG=120 timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_defer$i.log 2> $O/bench_defer$i.err < /dev/null
done
