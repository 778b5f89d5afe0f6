#!/bin/bash
# split-bf16 fc1 of the PPO epochs: parity on the GPU, then timing against the fp32 contraction (same box)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3o; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -m gpu -x -q -k "fc1_split or ppo" 2>&1 | tail -5
timeout 300 python scripts/microbench.py ppo_train 2>&1 | grep -v amdgpu.ids | tee $O/mb.txt
IPLAN_PPO_FC1_FP32=1 timeout 300 python scripts/microbench.py ppo_train 2>&1 | grep -v amdgpu.ids | sed 's/^/fp32 /' | tee -a $O/mb.txt
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
IPLAN_PPO_FC1_FP32=1 IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_fp32.json 2> $O/bench_fp32.err; cut -c1-200 $O/bench_fp32.json
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p" -o ppo -- python "$R/scripts/microbench.py" ppo_train > /dev/null 2> "$R/$O/prof.err" < /dev/null )
f=$(find $O/p -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-160; cp $f $O/ppo_train_kernel_stats.csv; rm -rf $O/p
