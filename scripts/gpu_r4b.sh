#!/bin/bash
# round 4, call b: A/B on one box -- behaviour encoder split-bf16 vs fp32 (IPLAN_ENC_FP32=1), behaviour learning enqueued first
# (IPLAN_BEH_FIRST=1), PPO fc1 form at a rank's 2 880 rows (IPLAN_PPO_FC1_FP32=1).   outputs -> gpurun_out/r4b/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r4b; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "behavior or ppo_train_config3 or deferred" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for rep in 1 2; do
  for v in base encfp32; do
    echo "== $v" >> $O/mb.txt
    if [ $v = encfp32 ]; then export IPLAN_ENC_FP32=1; else unset IPLAN_ENC_FP32; fi
    timeout 200 python scripts/microbench.py behavior_learn 2>&1 | grep -v amdgpu.ids >> $O/mb.txt
  done
done
unset IPLAN_ENC_FP32
bl() { IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],2), round(d['value']))"; }
for rep in 1 2; do
  echo "base       $(bl)" >> $O/cycle.txt
  echo "encfp32    $(IPLAN_ENC_FP32=1 bl)" >> $O/cycle.txt
  echo "behfirst   $(IPLAN_BEH_FIRST=1 bl)" >> $O/cycle.txt
  echo "runahead   $(IPLAN_RUN_AHEAD=1 bl)" >> $O/cycle.txt
  echo "first+ahead $(IPLAN_BEH_FIRST=1 IPLAN_RUN_AHEAD=1 bl)" >> $O/cycle.txt
done
for rep in 1 2; do
  echo "rank8 base     $(bl --scaling strong --emulate-rank-of 8)" >> $O/cycle.txt
  echo "rank8 fc1fp32  $(IPLAN_PPO_FC1_FP32=1 bl --scaling strong --emulate-rank-of 8)" >> $O/cycle.txt
  echo "rank8 behfirst $(IPLAN_BEH_FIRST=1 bl --scaling strong --emulate-rank-of 8)" >> $O/cycle.txt
  echo "rank8 fc1fp32+behfirst $(IPLAN_BEH_FIRST=1 IPLAN_PPO_FC1_FP32=1 bl --scaling strong --emulate-rank-of 8)" >> $O/cycle.txt
done
cat $O/mb.txt $O/cycle.txt; tail -3 $O/pytest_gpu.log
