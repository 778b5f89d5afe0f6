#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for v in "IPLAN_NO_DEFER_DECODER=1" "IPLAN_NO_CU_MASK=1" "IPLAN_DEFER_CUS=64" "IPLAN_DEFER_CUS=96" "IPLAN_NO_DEFER_DECODER=1" "IPLAN_NO_CU_MASK=1" "IPLAN_DEFER_CUS=64"; do
echo "== $v"; env $v IPLAN_BENCH_WATCHDOG=120 timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null < /dev/null | cut -c75-190
done > $O/defer_sweep.txt 2>&1
