#!/bin/bash
# Round-6 probe call 1: GPU suite on the round's host-side changes (launcher, DP recipe, grouped fp32 fc1 accumulation, conditioning-aware
# tolerance), the driver-shaped bench line through bench.py's own launcher, and same-box A/Bs of knobs the review asked to re-measure
# (HIP-graph replay of the one-launch-per-step rollout; the deferred decoder update on a CU-masked stream).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6p1; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --gpus 1 > $O/bench_line.json 2> $O/bench.err < /dev/null; echo "bench rc=$?" >> $O/bench.err
B="python bench.py --in-process --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
for rep in 1 2; do
for v in base graph cus48 cus64 cus96 behfirst; do
  unset IPLAN_ROLLOUT_GRAPH IPLAN_DEFER_CUS IPLAN_BEH_FIRST
  case $v in graph) export IPLAN_ROLLOUT_GRAPH=1;; cus48) export IPLAN_DEFER_CUS=48;; cus64) export IPLAN_DEFER_CUS=64;; cus96) export IPLAN_DEFER_CUS=96;; behfirst) export IPLAN_BEH_FIRST=1;; esac
  echo "== $v" >> $O/ab.txt
  IPLAN_BENCH_WATCHDOG=300 timeout 400 $B 2>> $O/ab.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('ms_per_step %.2f value %.0f fused_us %.1f' % (d['ms_per_step'], d['value'], r['us_per_launch']), 'coverage', r.get('rows_cover_kernel_time', {}).get('covered_frac'))
" >> $O/ab.txt
done; done
unset IPLAN_ROLLOUT_GRAPH IPLAN_DEFER_CUS IPLAN_BEH_FIRST
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --scaling strong --emulate-rank-of 8 --no-cpu-baseline > $O/bench_strong_rank_of_8_projection.json 2> $O/bench_proj.err < /dev/null
cat $O/ab.txt; tail -3 $O/pytest_gpu.log; cut -c1-300 $O/bench_line.json; tail -2 $O/bench.err
