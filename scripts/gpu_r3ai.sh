#!/bin/bash
# deferred decoder update (default) vs the in-line window-range wgrads (IPLAN_NO_DEFER_DECODER=1) on the final build
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3ai; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2 3; do
for v in defer inline; do
unset IPLAN_NO_DEFER_DECODER; [ $v = inline ] && export IPLAN_NO_DEFER_DECODER=1
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> $O/bench_${v}_$rep.err > $O/bench_${v}_$rep.json; echo "$v $(grep -o 'ms_per_step[^,]*' $O/bench_${v}_$rep.json)"
done; done
