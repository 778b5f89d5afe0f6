#!/bin/bash
# cycle time with the deferred decoder update on a plain side stream (0) or on a stream masked to k CUs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/defer; mkdir -p $O; : > $O/lines.txt
for k in 0 64 96 128 0; do
  echo "== IPLAN_DEFER_CUS=$k" >> $O/lines.txt
  IPLAN_DEFER_CUS=$k IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'])" >> $O/lines.txt
done
cat $O/lines.txt
