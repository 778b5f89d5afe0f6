#!/bin/bash
# A/B (same box): decoder / encoder BPTT load scheduling.  old = round-2b kernels, decbf = branch-free decoder loads only,
# prefetch = raw prefetch + consumer-side masking in both BPTT kernels.  Then tests + bench on the new library.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/abl; O=gpurun_out; export TMPDIR=/tmp
run() { # tag lib [env]
  ( cd /tmp && env ${3:-X=1} IPLAN_BEH_SERIAL=1 IPLAN_HIP_LIB=$2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/abl/$1" -o mb -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/abl/$1.log" 2>&1 )
  echo "== $1 (serial)"; grep -E "beh_dec|beh_enc|wgrad" "$O/abl/$1/mb_kernel_stats.csv" | awk -F, '{printf "%s calls %s avg_us %.1f\n",$1,$2,$4/1000}'
  grep behavior_learn "$O/abl/$1.log"
}
run old $R/build/abl/lib_syncthreads.so > $O/abl_summary.txt
run old_kernelsum $R/build/abl/lib_syncthreads.so IPLAN_BEH_KERNEL_WINSUM=1 >> $O/abl_summary.txt
run decbf $R/build/abl/lib_decbf.so >> $O/abl_summary.txt
run prefetch $R/iplan_amd/libiplan_hip.so >> $O/abl_summary.txt
for i in 1 2; do
IPLAN_HIP_LIB=$R/build/abl/lib_syncthreads.so timeout 200 python scripts/microbench.py behavior_learn prediction_learn > $O/ab_old$i.log 2>&1
timeout 200 python scripts/microbench.py behavior_learn prediction_learn > $O/ab_new$i.log 2>&1
done
grep -H "behavior_learn\|prediction_learn" $O/ab_*.log >> $O/abl_summary.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.log 2> $O/bench.err
IPLAN_HIP_LIB=$R/build/abl/lib_syncthreads.so IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_old.log 2> $O/bench_old.err
rm -rf $O/abl/*/
