#!/bin/bash
# Round-6 probe call 4: the cycle's time depends on which streams share a hardware queue (call 3: 264 ... 283 ms).  Does the explicit
# order train() -> Behavior_policy.learn in buffer-full cycles (harness.cycle, IPLAN_TRAIN_ORDER) remove the dependence?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6p4; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
A="--gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
for rep in 1 2; do
for q in late early_q4 early_q6 early_q8 early_q12; do
for ord in concurrent auto; do
  unset IPLAN_BENCH_PG_EARLY GPU_MAX_HW_QUEUES
  export IPLAN_TRAIN_ORDER=$ord
  case $q in late) ;; early_q4) export IPLAN_BENCH_PG_EARLY=1;; early_q6) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=6;;
            early_q8) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=8;; early_q12) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=12;; esac
  echo "== $q $ord" >> $O/ab.txt
  IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py $A 2>> $O/ab.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('ms_per_step %.2f value %.0f fused_us %.1f' % (d['ms_per_step'], d['value'], r['us_per_launch']))
" >> $O/ab.txt
done; done; done
unset IPLAN_BENCH_PG_EARLY GPU_MAX_HW_QUEUES IPLAN_TRAIN_ORDER
for ord in concurrent serial; do
  echo "== rank-of-8 $ord" >> $O/ab.txt
  IPLAN_TRAIN_ORDER=$ord IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --scaling strong --emulate-rank-of 8 --no-cpu-baseline --no-extras --steps 12 --warmup 3 2>> $O/ab.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('ms_per_step %.2f' % d['ms_per_step'])
" >> $O/ab.txt
  echo "== strong n1 $ord" >> $O/ab.txt
  IPLAN_TRAIN_ORDER=$ord IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --in-process --scaling strong --no-cpu-baseline --no-extras --steps 4 --warmup 1 2>> $O/ab.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('ms_per_step %.2f' % d['ms_per_step'])
" >> $O/ab.txt
done
paste - - < $O/ab.txt
