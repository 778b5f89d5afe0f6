#!/bin/bash
# distribution of the driver-shaped line over 30 fresh processes on one box (the rare slow stream -> queue assignment: does the check
# after the first cycle catch it?)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6dist; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for i in $(seq 1 30); do
  IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>> $O/err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%.2f' % d['ms_per_step'], d['launcher']['hardware_queue_probe']['streams_replaced_after_first_cycle'])
" >> $O/dist.txt
done
sort -n $O/dist.txt | tr '\n' ';'
