#!/bin/bash
# the evidence call's pieces that changed after it (host-side only: kernel sources and their hash are those of series r04a):
# bench line with the scenes' span, kernel statistics / traces of the bench command without the config-5 / config-4 extras
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; S=${SERIES:-r04a}; O=gpurun_out/$S; mkdir -p $O; export TMPDIR=/tmp
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py > $O/bench_line.json 2> $O/bench.err < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --scaling strong --emulate-rank-of 8 --no-cpu-baseline > $O/bench_strong_rank_of_8_projection.json 2> $O/bench_proj.err < /dev/null
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_cycle" -o cyc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$R/$O/bench_under_rocprof.json" 2> "$R/$O/bench_under_rocprof.err" < /dev/null )
find $O/prof_cycle -name "*kernel_stats.csv" -exec cp {} $O/full_cycle_kernel_stats.csv \;
f=$(find $O/prof_cycle -name "*kernel_trace.csv" | head -1)
python scripts/trace_busy.py $f > $O/cycle_trace_busy.txt 2>&1; python scripts/trace_learn.py $f > $O/cycle_trace_learn_phase.txt 2>&1
rm -rf $O/prof_cycle
cut -c1-600 $O/bench_line.json; cat $O/cycle_trace_busy.txt; head -4 $O/full_cycle_kernel_stats.csv | cut -c1-150; head -3 $O/cycle_trace_learn_phase.txt
