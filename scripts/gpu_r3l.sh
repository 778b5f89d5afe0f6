#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3l; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_rollout.py -m gpu -q 2>&1 | tail -2
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err; cut -c1-330 $O/bench.json
IPLAN_FRESH_BATCH=1 IPLAN_BEH_EQUAL_PIECES=1 IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_old_prologue.json 2> $O/bench_old.err; cut -c1-330 $O/bench_old_prologue.json
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p" -o cyc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/$O/bench_traced.json" 2> "$R/$O/bench_traced.err" < /dev/null )
f=$(find $O/p -name "*kernel_trace.csv" | head -1)
python scripts/trace_busy.py $f > $O/cycle_trace_busy.txt; tail -12 $O/cycle_trace_busy.txt
python scripts/trace_learn.py $f > $O/cycle_trace_learn_phase.txt; head -48 $O/cycle_trace_learn_phase.txt | cut -c1-330
rm -rf $O/p
