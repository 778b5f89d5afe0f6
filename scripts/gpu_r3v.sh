#!/bin/bash
# wide wgrad jobs on the bf16 matrix cores (default) vs fp32 MFMA (build/abl/lib_wfp32.so): parity, piece timings, cycle
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3v; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
for rep in 1 2; do
for v in bf16 wfp32; do
lib=$R/iplan_amd/libiplan_hip.so; [ $v != bf16 ] && lib=$R/build/abl/lib_$v.so
IPLAN_HIP_LIB=$lib timeout 300 python scripts/microbench.py behavior_learn ppo_train 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" | tee -a $O/mb.txt
IPLAN_HIP_LIB=$lib IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err; cut -c1-200 $O/bench_${v}_$rep.json
done; done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p" -o cyc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/$O/bench_traced.json" 2> "$R/$O/bench_traced.err" < /dev/null )
f=$(find $O/p -name "*kernel_trace.csv" | head -1)
python scripts/trace_busy.py $f > $O/cycle_trace_busy.txt; tail -12 $O/cycle_trace_busy.txt
python scripts/trace_learn.py $f > $O/cycle_trace_learn_phase.txt; head -52 $O/cycle_trace_learn_phase.txt | cut -c1-150
rm -rf $O/p
