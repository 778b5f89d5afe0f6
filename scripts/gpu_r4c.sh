#!/bin/bash
# round 4, call c: decoder BPTT second form (default) vs first form (IPLAN_DEC_BWD_V1=1) on one box: parity tests, learn alone, cycle
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r4c; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "behavior or deferred or properties or env_independence" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
for rep in 1 2; do
  for v in base bwdv1; do
    echo "== $v" >> $O/mb.txt
    if [ $v = bwdv1 ]; then export IPLAN_DEC_BWD_V1=1; else unset IPLAN_DEC_BWD_V1; fi
    timeout 200 python scripts/microbench.py behavior_learn 2>&1 | grep -v amdgpu.ids >> $O/mb.txt
  done
done
unset IPLAN_DEC_BWD_V1
bl() { IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],2), round(d['value']), [ (r['kernel'][:18], round(r['us_per_launch'])) for r in d['roofline_others'][:4]])"; }
for rep in 1 2; do
  echo "base       $(bl)" >> $O/cycle.txt
  echo "bwdv1      $(IPLAN_DEC_BWD_V1=1 bl)" >> $O/cycle.txt
done
# serial kernel times of one learn (kernel trace of the microbench piece)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p1" -o mb -- python "$R/scripts/microbench.py" behavior_learn > /dev/null 2>&1 )
find $O/p1 -name "*kernel_stats.csv" -exec cp {} $O/behaviour_learn_kernel_stats.csv \;
rm -rf $O/p1
cat $O/mb.txt $O/cycle.txt; head -12 $O/behaviour_learn_kernel_stats.csv | cut -c1-200
