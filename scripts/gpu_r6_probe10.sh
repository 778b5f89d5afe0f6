#!/bin/bash
# Round-6 probe call 6: streams that must overlap the main stream are PROBED for a hardware queue of their own (streams.distinct_stream).
# Is the cycle now independent of the stream -> queue accident?  (no RCCL group / RCCL alive at 4, 5, 6, 8 queues; probe on / off)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6p10; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
A="--gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
for rep in 1 2; do
for q in late late early_q6; do
for pr in full full_decown noprobe; do
  unset IPLAN_BENCH_PG_EARLY GPU_MAX_HW_QUEUES IPLAN_NO_QUEUE_PROBE
  unset IPLAN_QUEUE_PROBE; [ $pr = noprobe ] && export IPLAN_QUEUE_PROBE=0; [ $pr != noprobe ] && export IPLAN_QUEUE_PROBE=full; unset IPLAN_DEC_OWN_STREAM; [ $pr = full_decown ] && export IPLAN_DEC_OWN_STREAM=1
  case $q in late) ;; early_q4) export IPLAN_BENCH_PG_EARLY=1;; early_q5) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=5;; early_q6) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=6;;
            early_q8) export IPLAN_BENCH_PG_EARLY=1 GPU_MAX_HW_QUEUES=8;; esac
  echo "== $q $pr" >> $O/ab.txt
  IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py $A 2>> $O/ab.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('ms_per_step %.2f value %.0f fused_us %.1f' % (d['ms_per_step'], d['value'], r['us_per_launch']))
" >> $O/ab.txt
done; done; done
unset IPLAN_BENCH_PG_EARLY GPU_MAX_HW_QUEUES IPLAN_NO_QUEUE_PROBE
for v in full noprobe full noprobe; do
  unset IPLAN_QUEUE_PROBE; [ $v = noprobe ] && export IPLAN_QUEUE_PROBE=0; [ $v != noprobe ] && export IPLAN_QUEUE_PROBE=full; unset IPLAN_PPO_OWN_STREAM; [ $v = full_own ] && export IPLAN_PPO_OWN_STREAM=1
  echo "== rank-of-8 $v" >> $O/ab.txt
  IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --scaling strong --emulate-rank-of 8 --no-cpu-baseline --no-extras --steps 30 --warmup 3 2>> $O/ab.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('ms_per_step %.2f' % d['ms_per_step'])
" >> $O/ab.txt
done
unset IPLAN_QUEUE_PROBE
paste - - < $O/ab.txt

