#!/bin/bash
# column-grouped decoder records (this tree) vs the previous commit's chain-major records (build/old_tree: git archive HEAD + make)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3z; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
for rep in 1; do
for v in new old; do
T=$R; [ $v = old ] && T=$R/build/old_tree
( cd $T && timeout 300 python scripts/microbench.py behavior_learn 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" ) | tee -a $O/mb.txt
( cd $T && IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> $R/$O/bench_${v}_$rep.err ) > $O/bench_${v}_$rep.json; cut -c1-200 $O/bench_${v}_$rep.json
done; done
for v in new old; do
T=$R; [ $v = old ] && T=$R/build/old_tree
( cd /tmp && IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p" -o beh -- python "$T/scripts/microbench.py" behavior_learn > /dev/null 2>&1 < /dev/null )
f=$(find $O/p -name "*kernel_stats.csv" | head -1); echo "== $v (serial behaviour learn)"; head -9 $f | cut -c1-140; cp $f $O/behaviour_serial_kernel_stats_$v.csv; rm -rf $O/p
done
