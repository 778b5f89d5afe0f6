#!/bin/bash
# the deferred decoder update recorded and run in line at the head of the NEXT learn phase (IPLAN_DEFER_NEXT=1) vs beside the next rollout
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3ah; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2 3; do
for v in base next; do
unset IPLAN_DEFER_NEXT; [ $v = next ] && export IPLAN_DEFER_NEXT=1
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> $O/bench_${v}_$rep.err > $O/bench_${v}_$rep.json; echo "$v $(grep -o 'ms_per_step[^,]*' $O/bench_${v}_$rep.json)"
done; done
export IPLAN_DEFER_NEXT=1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p" -o cyc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 < /dev/null )
f=$(find $O/p -name "*kernel_trace.csv" | head -1); python scripts/trace_busy.py $f | tail -9; python scripts/trace_learn.py $f > $O/cycle_trace_learn_phase.txt; head -16 $O/cycle_trace_learn_phase.txt | cut -c1-110; rm -rf $O/p
