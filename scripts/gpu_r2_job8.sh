#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/abl; O=gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
run() { # tag lib
  ( cd /tmp && IPLAN_BEH_SERIAL=1 IPLAN_HIP_LIB=$2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/abl/$1" -o mb -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/abl/$1.log" 2>&1 )
  echo "== $1 (serial)"; grep -E "beh_dec|beh_enc" "$O/abl/$1/mb_kernel_stats.csv" | awk -F, '{printf "%s calls %s avg_us %.1f\n",$1,$2,$4/1000}'
}
run syncthreads $R/build/abl/lib_syncthreads.so > $O/abl_summary.txt
run ldsbarrier $R/iplan_amd/libiplan_hip.so >> $O/abl_summary.txt
for i in 1 2; do
IPLAN_HIP_LIB=$R/build/abl/lib_syncthreads.so timeout 200 python scripts/microbench.py behavior_learn prediction_learn > $O/ab_sync$i.log 2>&1
timeout 200 python scripts/microbench.py behavior_learn prediction_learn > $O/ab_lds$i.log 2>&1
done
IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.log 2> $O/bench.err
IPLAN_HIP_LIB=$R/build/abl/lib_syncthreads.so IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_sync.log 2> $O/bench_sync.err
rm -rf $O/abl/*/
