#!/bin/bash
# Round-6 probe call 3: with a live RCCL group (every N > 1 rank), which GPU_MAX_HW_QUEUES gives the cycle its un-aliased streams back?
# + the launcher's N = 1 fallback test.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6p3; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
A="--gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
for rep in 1 2; do
for v in late early_q4 early_q5 early_q6 early_q8 early_q12; do
  unset IPLAN_BENCH_PG_EARLY GPU_MAX_HW_QUEUES
  case $v in late) ;; early_q4) export IPLAN_BENCH_PG_EARLY=1;; early_q5) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=5;; early_q6) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=6;;
            early_q8) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=8;; early_q12) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=12;; esac
  echo "== $v" >> $O/ab.txt
  IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py $A 2>> $O/ab.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('ms_per_step %.2f value %.0f fused_us %.1f' % (d['ms_per_step'], d['value'], r['us_per_launch']))
" >> $O/ab.txt
done; done
unset IPLAN_BENCH_PG_EARLY GPU_MAX_HW_QUEUES
timeout 600 python -m pytest tests/test_bench_launcher.py -m gpu -q > $O/pytest_launcher.log 2>&1; echo "rc=$?" >> $O/pytest_launcher.log
cat $O/ab.txt; tail -3 $O/pytest_launcher.log
