#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/fuse; mkdir -p $O; : > $O/lines.txt
timeout 300 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_parity_fullsize.py -m gpu -q -k "rollout" 2>&1 | tail -2 >> $O/lines.txt
for k in 1 "" 1 ""; do
  echo "== IPLAN_NO_FUSE_ENC=$k" >> $O/lines.txt
  IPLAN_NO_FUSE_ENC=$k IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'])" >> $O/lines.txt
  IPLAN_NO_FUSE_ENC=$k timeout 100 python scripts/microbench.py rollout 2>&1 | grep rollout >> $O/lines.txt
done
cat $O/lines.txt
