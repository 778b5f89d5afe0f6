#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6pieces; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
for v in "4 6" "3 6" "6 6" "4 4" "4 8" "4 10" "6 8"; do
  set -- $v
  echo "== fwd $1 bwd $2" >> $O/ab.txt
  IPLAN_BEH_PIECES_FWD=$1 IPLAN_BEH_PIECES_BWD=$2 IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --in-process --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>> $O/err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('ms_per_step %.2f' % d['ms_per_step'])
" >> $O/ab.txt
done; done
paste - - < $O/ab.txt
