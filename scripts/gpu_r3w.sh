#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3w; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity_fullsize.py -m gpu -x -q 2>&1 | tail -2
for v in bf16 wfp32; do
lib=$R/iplan_amd/libiplan_hip.so; [ $v != bf16 ] && lib=$R/build/abl/lib_$v.so
IPLAN_HIP_LIB=$lib timeout 300 python scripts/microbench.py behavior_learn ppo_train 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" | tee -a $O/mb.txt
IPLAN_HIP_LIB=$lib IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_${v}.json 2> $O/bench_${v}.err; cut -c1-200 $O/bench_${v}.json
IPLAN_HIP_LIB=$lib timeout 400 python scripts/cfg5_bench.py 2>&1 | grep -v amdgpu.ids | tail -4 | sed "s/^/$v /" | tee -a $O/cfg5.txt
done
