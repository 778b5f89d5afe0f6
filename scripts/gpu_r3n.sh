#!/bin/bash
# A/B of the wide wgrad shape: build/abl/lib_{base,to6,to6rb2}.so -- behaviour learn microbench, kernel stats, bench cycle
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3n; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for rep in 1; do
for lib in base to6r1 to6r1rb2; do
  echo "== $lib mb" | tee -a $O/mb.txt
  IPLAN_HIP_LIB=$R/build/abl/lib_$lib.so timeout 200 python scripts/microbench.py behavior_learn prediction_learn ppo_train 2>&1 | grep -v amdgpu.ids | tee -a $O/mb.txt
  IPLAN_HIP_LIB=$R/build/abl/lib_$lib.so IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_${lib}_$rep.json 2> $O/bench_${lib}_$rep.err; cut -c1-200 $O/bench_${lib}_$rep.json
done; done
for lib in to6r1 to6r1rb2; do
( cd /tmp && IPLAN_HIP_LIB=$R/build/abl/lib_$lib.so timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p_$lib" -o cyc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/$O/bench_traced_$lib.json" 2> "$R/$O/bench_traced_$lib.err" < /dev/null )
f=$(find $O/p_$lib -name "*kernel_trace.csv" | head -1)
python scripts/trace_busy.py $f > $O/cycle_trace_busy_$lib.txt; tail -12 $O/cycle_trace_busy_$lib.txt
python scripts/trace_learn.py $f > $O/cycle_trace_learn_phase_$lib.txt; head -44 $O/cycle_trace_learn_phase_$lib.txt | cut -c1-250
rm -rf $O/p_$lib
done
