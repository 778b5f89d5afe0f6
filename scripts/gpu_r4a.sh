#!/bin/bash
# round 4, call a: new parity tests, bench line with config5 / config4_n1, rank-of-8 projection split, kernel traces of the learn
# phase head and of a PPO epoch at 22 950 and 2 880 rows.   outputs -> gpurun_out/r4a/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r4a; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py > $O/bench_line.json 2> $O/bench.err < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --scaling strong --emulate-rank-of 8 --no-cpu-baseline > $O/bench_proj.json 2> $O/bench_proj.err < /dev/null
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p1" -o cyc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$R/$O/bench_traced.json" 2> "$R/$O/bench_traced.err" < /dev/null )
f=$(find $O/p1 -name "*kernel_trace.csv" | head -1)
python scripts/trace_busy.py $f > $O/cycle_trace_busy.txt 2>&1; python scripts/trace_learn.py $f > $O/cycle_trace_learn_phase.txt 2>&1
python scripts/trace_window.py $f 3.0 > $O/trace_window_cfg3.txt 2>&1
rm -rf $O/p1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p2" -o cyc -- python "$R/bench.py" --scaling strong --emulate-rank-of 8 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > "$R/$O/bench_proj_traced.json" 2> "$R/$O/bench_proj_traced.err" < /dev/null )
f=$(find $O/p2 -name "*kernel_trace.csv" | head -1)
python scripts/trace_window.py $f 3.0 > $O/trace_window_rank_of_8.txt 2>&1
rm -rf $O/p2
timeout 200 python scripts/dev/host_pace.py > $O/host_pace.txt 2>&1
ls -la $O; tail -5 $O/pytest_gpu.log; cut -c1-300 $O/bench_line.json
