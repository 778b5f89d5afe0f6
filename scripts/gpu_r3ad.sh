#!/bin/bash
# behaviour learner's data-movement head enqueued before the side learners (this tree) vs after them (previous commit, build/old_tree)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3ad; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2 3; do
for v in new old; do
T=$R; [ $v = old ] && T=$R/build/old_tree
( cd $T && IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> $R/$O/bench_${v}_$rep.err ) > $O/bench_${v}_$rep.json; echo "$v $(grep -o 'ms_per_step[^,]*' $O/bench_${v}_$rep.json)"
done; done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p" -o cyc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 < /dev/null )
f=$(find $O/p -name "*kernel_trace.csv" | head -1); python scripts/trace_learn.py $f > $O/cycle_trace_learn_phase.txt; head -12 $O/cycle_trace_learn_phase.txt | cut -c1-110; rm -rf $O/p
