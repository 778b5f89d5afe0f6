#!/bin/bash
# host-side knobs re-measured on the final build: rollout graph replay, host run-ahead
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3ag; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
for v in base graph ahead both; do
unset IPLAN_ROLLOUT_GRAPH IPLAN_RUN_AHEAD
case $v in graph) export IPLAN_ROLLOUT_GRAPH=1;; ahead) export IPLAN_RUN_AHEAD=1;; both) export IPLAN_ROLLOUT_GRAPH=1 IPLAN_RUN_AHEAD=1;; esac
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> $O/bench_${v}_$rep.err > $O/bench_${v}_$rep.json; echo "$v $(grep -o 'ms_per_step[^,]*' $O/bench_${v}_$rep.json)"
done; done
