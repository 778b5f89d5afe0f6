#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/trace; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p" -o cyc -- python "$R/bench.py" --in-process --steps 2 --warmup 1 --no-cpu-baseline > "$R/$O/bench.json" 2> "$R/$O/bench.err" < /dev/null )
f=$(find $O/p -name "*kernel_trace.csv" | head -1)
python scripts/trace_busy.py $f | tee $O/busy.txt
python scripts/trace_learn.py $f | tee $O/learn.txt
rm -rf $O/p
