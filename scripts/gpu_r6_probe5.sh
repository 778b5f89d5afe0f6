#!/bin/bash
# Round-6 probe call 5: kernel traces of the cycle under three stream -> hardware-queue mappings (no RCCL group / RCCL alive at 4 and 6
# queues): queue of every kernel family, the learn phase's timeline, rollout / learn / train phase lengths.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6p5; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for v in late early_q4 early_q6 early_q5; do
  unset IPLAN_BENCH_PG_EARLY GPU_MAX_HW_QUEUES
  case $v in late) ;; early_q4) export IPLAN_BENCH_PG_EARLY=1;; early_q6) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=6;; early_q5) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=5;; esac
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p_$v" -o cyc -- python "$R/bench.py" --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$R/$O/bench_$v.json" 2> "$R/$O/bench_$v.err" < /dev/null )
  f=$(ls -S $(find $O/p_$v -name "*kernel_trace.csv") | head -1)
  echo "== $v  $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.json)" > $O/trace_$v.txt
  python scripts/dev/queue_map.py $f >> $O/trace_$v.txt 2>&1
  python scripts/trace_busy.py $f >> $O/trace_$v.txt 2>&1
  python scripts/trace_learn.py $f >> $O/trace_$v.txt 2>&1
  rm -rf $O/p_$v
done
head -60 $O/trace_late.txt
