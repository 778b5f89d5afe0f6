#!/bin/bash
# the deferred decoder update on a CU-masked stream (IPLAN_DEFER_CUS), re-measured with the 2.8 ms bf16 wide wgrad
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3ab; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for cus in 0 64 96 80 0 64; do
IPLAN_DEFER_CUS=$cus IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> $O/bench_$cus.err > $O/bench_$cus.json; echo "cus=$cus $(cut -c1-190 $O/bench_$cus.json | grep -o 'ms_per_step[^,]*')"
done
