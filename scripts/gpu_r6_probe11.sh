#!/bin/bash
# Round-6 probe call 11: IPLAN_QUEUE_PROBE=verify (creation order, the critical pairs probed AFTER the first cycle, violators replaced)
# next to full and unprobed, without RCCL (x4) and with a live group at 4 / 6 queues.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6p11; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
A="--gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
for rep in 1 2; do
for q in late late early_q4 early_q6; do
for pr in verify full 0; do
  unset IPLAN_BENCH_PG_EARLY GPU_MAX_HW_QUEUES
  export IPLAN_QUEUE_PROBE=$pr
  case $q in late) ;; early_q4) export IPLAN_BENCH_PG_EARLY=1;; early_q6) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=6;; esac
  echo "== $q $pr" >> $O/ab.txt
  IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py $A 2>> $O/ab.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('ms_per_step %.2f value %.0f fused_us %.1f' % (d['ms_per_step'], d['value'], r['us_per_launch']), d['launcher']['hardware_queue_probe'])
" >> $O/ab.txt
done; done; done
paste - - < $O/ab.txt
