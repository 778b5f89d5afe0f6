#!/bin/bash
# window-range counts of the behaviour forward / BPTT pipelines, re-tuned for the second-form decoder forward
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3af; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for cfg in "4 6" "3 6" "2 6" "6 6" "4 4" "4 8" "3 5" "4 6"; do
set -- $cfg
IPLAN_BEH_PIECES_FWD=$1 IPLAN_BEH_PIECES_BWD=$2 IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> $O/bench_$1_$2.err > $O/bench_$1_$2.json; echo "fwd=$1 bwd=$2 $(grep -o 'ms_per_step[^,]*' $O/bench_$1_$2.json)"
done
