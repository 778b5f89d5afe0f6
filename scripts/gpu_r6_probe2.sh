#!/bin/bash
# Round-6 probe call 2: does the 1-rank RCCL group cost the single-GPU line 3 %?  (spawned + group created late / early, in-process);
# the tests touched since call 1; every PPO optimiser step against the oracle once (scripts/ppo_all_steps_check.py).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6p2; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
A="--steps 6 --warmup 2 --no-cpu-baseline --no-extras"
for rep in 1 2; do
for v in inproc spawn_late spawn_early; do
  unset IPLAN_BENCH_PG_EARLY
  case $v in inproc) C="python bench.py --in-process $A";; spawn_late) C="python bench.py --gpus 1 $A";; spawn_early) export IPLAN_BENCH_PG_EARLY=1; C="python bench.py --gpus 1 $A";; esac
  echo "== $v" >> $O/ab.txt
  IPLAN_BENCH_WATCHDOG=300 timeout 400 $C 2>> $O/ab.err > $O/line_$v.json
  echo "stdout lines: $(grep -c . $O/line_$v.json)" >> $O/ab.txt
  python -c "
import sys, json
for ln in open('$O/line_$v.json'):
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('ms_per_step %.2f value %.0f fused_us %.1f' % (d['ms_per_step'], d['value'], r['us_per_launch']), d['launcher'])
" >> $O/ab.txt
done; done
unset IPLAN_BENCH_PG_EARLY
timeout 900 python -m pytest tests/test_bench_launcher.py "tests/test_gpu_parity_fullsize.py::test_ppo_loss_switches_vs_oracle" "tests/test_gpu_parity_fullsize.py::test_fc1_split_vs_fp32_gpu" "tests/test_gpu_parity_fullsize.py::test_ppo_train_config3_15_epochs_vs_oracle" -m gpu -q -x > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
( time timeout 1500 python scripts/ppo_all_steps_check.py $O/ppo_all_steps_agent4.json 4 ) > $O/ppo_all_steps.log 2>&1; echo "rc=$?" >> $O/ppo_all_steps.log
cat $O/ab.txt; tail -4 $O/pytest_subset.log; tail -6 $O/ppo_all_steps.log | cut -c1-600
