#!/bin/bash
# Round-2 GPU call 2: quarter-split decoder kernels -- parity tests + behaviour-learn timing + kernel stats
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; O=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
rm -f $O/parity_errors.json
timeout 900 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python scripts/microbench.py behavior_learn rollout ppo_train > $O/microbench.log 2>&1
( cd /tmp && MB_PIECES="behavior_learn" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_mb" -o mb -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/prof_mb.log" 2>&1 )
python scripts/prof_summary.py $O/prof_mb/mb_kernel_stats.csv $O/prof_mb_summary.csv > /dev/null 2>&1
IPLAN_BENCH_WATCHDOG=200 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
timeout 200 python scripts/bench_runner.py > $O/bench_runner.log 2>&1
rm -rf $O/prof_mb/*/ 2>/dev/null
