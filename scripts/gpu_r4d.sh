#!/bin/bash
# round 4, call d: in-kernel thin decoder weight gradients (default) vs row gradients (IPLAN_DEC_THIN_ROWS=1) vs BPTT first form;
# small-batch fc1 forward shape at a rank's 2 880 rows; serial per-kernel times of one learn.   outputs -> gpurun_out/r4d/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r4d; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "behavior or deferred or properties or env_independence or fc1_split or dp_gpu" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
bl() { IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],2), round(d['value']), [ (r['kernel'][:18], round(r['us_per_launch'])) for r in d['roofline_others'][:4]])"; }
for rep in 1 2; do
  echo "base       $(bl)" >> $O/cycle.txt
  echo "thinrows   $(IPLAN_DEC_THIN_ROWS=1 bl)" >> $O/cycle.txt
  echo "bwdv1      $(IPLAN_DEC_BWD_V1=1 bl)" >> $O/cycle.txt
done
for rep in 1 2; do
  echo "rank8 base        $(bl --scaling strong --emulate-rank-of 8)" >> $O/cycle.txt
  echo "rank8 fullshape   $(IPLAN_AC_SPLIT_SHAPE=full bl --scaling strong --emulate-rank-of 8)" >> $O/cycle.txt
done
for v in base thinrows bwdv1; do
  unset IPLAN_DEC_THIN_ROWS IPLAN_DEC_BWD_V1
  [ $v = thinrows ] && export IPLAN_DEC_THIN_ROWS=1
  [ $v = bwdv1 ] && export IPLAN_DEC_BWD_V1=1
  ( cd /tmp && IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p_$v" -o beh -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/serial_$v.log" 2>&1 < /dev/null )
  echo "== serial $v" >> $O/serial.txt; find $O/p_$v -name "*kernel_stats.csv" -exec grep -E "beh_dec|beh_enc|wgrad_" {} \; | awk -F, '{printf "%s calls %s avg_ns %s\n",$1,$2,$4}' >> $O/serial.txt; grep behavior_learn $O/serial_$v.log >> $O/serial.txt
  rm -rf $O/p_$v
done
unset IPLAN_DEC_THIN_ROWS IPLAN_DEC_BWD_V1
cat $O/cycle.txt $O/serial.txt
