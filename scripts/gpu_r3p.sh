#!/bin/bash
# split-bf16 fc1: parity, then 4-wave (default) vs 8-wave workgroups vs the fp32 contraction on one box
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3p; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -m gpu -x -q -k "fc1_split or ppo" 2>&1 | tail -5
for rep in 1 2; do
timeout 300 python scripts/microbench.py ppo_train 2>&1 | grep -v amdgpu.ids | sed 's/^/nw4  /' | tee -a $O/mb.txt
IPLAN_HIP_LIB=$R/build/abl/lib_nw8.so timeout 300 python scripts/microbench.py ppo_train 2>&1 | grep -v amdgpu.ids | sed 's/^/nw8  /' | tee -a $O/mb.txt
IPLAN_PPO_FC1_FP32=1 timeout 300 python scripts/microbench.py ppo_train 2>&1 | grep -v amdgpu.ids | sed 's/^/fp32 /' | tee -a $O/mb.txt
done
for v in nw4 nw8; do
lib=$R/iplan_amd/libiplan_hip.so; [ $v = nw8 ] && lib=$R/build/abl/lib_nw8.so
( cd /tmp && IPLAN_HIP_LIB=$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p" -o ppo -- python "$R/scripts/microbench.py" ppo_train > /dev/null 2> "$R/$O/prof.err" < /dev/null )
f=$(find $O/p -name "*kernel_stats.csv" | head -1); echo "== $v"; head -12 $f | cut -c1-150; cp $f $O/ppo_train_kernel_stats_$v.csv; rm -rf $O/p
done
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
