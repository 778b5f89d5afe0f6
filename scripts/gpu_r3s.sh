#!/bin/bash
# full GPU suite on the split-bf16 build, then 8-wave x 4-tile (default) vs 4-wave x 2-tile workgroups
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3s; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
for v in base nw4rt2; do
lib=$R/iplan_amd/libiplan_hip.so; [ $v != base ] && lib=$R/build/abl/lib_$v.so
( cd /tmp && IPLAN_HIP_LIB=$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p" -o ppo -- python "$R/scripts/microbench.py" ppo_train > "$R/$O/mb_$v.txt" 2> "$R/$O/prof.err" < /dev/null )
f=$(find $O/p -name "*kernel_stats.csv" | head -1); echo "== $v" | tee -a $O/abl.txt; grep ppo_train $O/mb_$v.txt; head -14 $f | cut -c1-130 | tee -a $O/abl.txt; rm -rf $O/p
done
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
