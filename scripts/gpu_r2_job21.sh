#!/bin/bash
# deferred decoder update on a 64-CU masked stream: tests, RCCL single-rank check, same-box A/B of the whole cycle
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $O/pytest_gpu.log
MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 200 python scripts/dp_single_rank_check.py > $O/dp_check.log 2>&1 < /dev/null; echo "dp rc=$?" >> $O/dp_check.log
for i in 1 2; do
IPLAN_NO_DEFER_DECODER=1 IPLAN_BENCH_WATCHDOG=120 timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_inline$i.log 2> $O/bench_inline$i.err < /dev/null
IPLAN_BENCH_WATCHDOG=120 timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_defer$i.log 2> $O/bench_defer$i.err < /dev/null
done
IPLAN_NO_CU_MASK=1 IPLAN_BENCH_WATCHDOG=120 timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_defer_nomask.log 2> $O/bench_defer_nomask.err < /dev/null
