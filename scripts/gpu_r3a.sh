#!/bin/bash
# Round-3 call a: GPU tests with the same-point PPO gradient check, the per-tensor PPO gradient error tables (default build and
# -DIPLAN_EXACT_GATES build), a baseline bench line of the round-2 kernels on this box.  Outputs -> gpurun_out/r3a/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3a; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.json
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
timeout 900 python scripts/ppo_grad_error_table.py --cases mb3x2,switches,cfg3_1,cfg3_2,cfg3_3 --json $O/ppo_grad_table_default.json > $O/ppo_grad_table_default.txt 2>&1
IPLAN_HIP_LIB=$R/build/abl/lib_exact.so timeout 900 python scripts/ppo_grad_error_table.py --cases mb3x2,switches,cfg3_1,cfg3_2,cfg3_3 --json $O/ppo_grad_table_exact_gates.json > $O/ppo_grad_table_exact_gates.txt 2>&1
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err
timeout 400 python scripts/microbench.py > $O/microbench.txt 2>&1
tail -3 $O/pytest_gpu.log; grep -h "^==" $O/ppo_grad_table_*.txt | cut -c1-400; cut -c1-300 $O/bench_line.json; cat $O/microbench.txt | grep -v amdgpu
