#!/bin/bash
# PPO train after: finalize rewrite, vectorised sqnorm, old_logp from epoch 0, critic wgrad on a side stream
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3u; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
for rep in 1 2; do
timeout 300 python scripts/microbench.py ppo_train 2>&1 | grep -v amdgpu.ids | sed 's/^/new    /' | tee -a $O/mb.txt
IPLAN_AC_WGRAD_SERIAL=1 timeout 300 python scripts/microbench.py ppo_train 2>&1 | grep -v amdgpu.ids | sed 's/^/serial /' | tee -a $O/mb.txt
done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p" -o ppo -- python "$R/scripts/microbench.py" ppo_train > /dev/null 2> "$R/$O/prof.err" < /dev/null )
f=$(find $O/p -name "*kernel_stats.csv" | head -1); head -24 $f | cut -c1-130; cp $f $O/ppo_train_kernel_stats.csv; rm -rf $O/p
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
