#!/bin/bash
# Round-3 call i: decoder forward second form -- GPU behaviour tests, serial per-kernel times (rocprofv3) and microbench, v1 vs v2
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3i; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "behavior or behaviour or deferred or fullsize or learners" > $O/pytest_beh.log 2>&1; echo "pytest rc=$?" >> $O/pytest_beh.log
tail -4 $O/pytest_beh.log | cut -c1-250
for v in v2 v1; do
  if [ $v = v1 ]; then export IPLAN_DEC_FWD_V1=1; else unset IPLAN_DEC_FWD_V1; fi
  ( cd /tmp && IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p_$v" -o beh -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/serial_$v.log" 2>&1 < /dev/null )
  echo "== $v serial"; find $O/p_$v -name "*kernel_stats.csv" -exec grep -E "beh_dec|beh_enc|wgrad_partial_kernel<12" {} \; | awk -F, '{printf "%s calls %s avg_ns %s\n",$1,$2,$4}'; grep behavior_learn $O/serial_$v.log
  rm -rf $O/p_$v
  timeout 300 python scripts/microbench.py behavior_learn rollout > $O/mb_$v.log 2>&1; grep -v amdgpu $O/mb_$v.log
done
unset IPLAN_DEC_FWD_V1
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_v2.json 2> $O/bench_v2.err; cut -c1-330 $O/bench_v2.json
IPLAN_DEC_FWD_V1=1 IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_v1.json 2> $O/bench_v1.err; cut -c1-330 $O/bench_v1.json
