#!/bin/bash
# Round-6 probe call 7: kernel traces with the probed stream assignment (no RCCL / RCCL alive, 4 queues) next to the creation-order one.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6p7; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for v in late_probe late_noprobe early_q4_probe; do
  unset IPLAN_BENCH_PG_EARLY GPU_MAX_HW_QUEUES IPLAN_NO_QUEUE_PROBE
  case $v in late_probe) ;; late_noprobe) export IPLAN_NO_QUEUE_PROBE=1;; early_q4_probe) export IPLAN_BENCH_PG_EARLY=1;; esac
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p_$v" -o cyc -- python "$R/bench.py" --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$R/$O/bench_$v.json" 2> "$R/$O/bench_$v.err" < /dev/null )
  f=$(ls -S $(find $O/p_$v -name "*kernel_trace.csv") | head -1)
  echo "== $v  $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.json)" > $O/trace_$v.txt
  python scripts/dev/queue_map.py $f >> $O/trace_$v.txt 2>&1
  python scripts/trace_busy.py $f >> $O/trace_$v.txt 2>&1
  python scripts/trace_learn.py $f >> $O/trace_$v.txt 2>&1
  rm -rf $O/p_$v
done
