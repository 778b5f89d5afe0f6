#!/bin/bash
# last-call checks on the final sources: the driver's own command line, and the forward tail's workgroup width at a rank's 2 880 rows
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/final; mkdir -p $O; export TMPDIR=/tmp
( time IPLAN_BENCH_WATCHDOG=280 timeout 290 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err ) 2> $O/bench_driver_like.time
cut -c1-230 $O/bench_driver_like.json; grep real $O/bench_driver_like.time
for w in 8 16; do
  IPLAN_AC_PRE_WAVES=$w IPLAN_BENCH_WATCHDOG=200 timeout 220 python bench.py --scaling strong --emulate-rank-of 8 --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('rank-of-8 step, forward tail $w waves:', round(d['ms_per_step'], 2), 'ms')" | tee -a $O/rank8_pre_waves.txt
done
