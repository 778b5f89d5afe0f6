#!/bin/bash
# decoder forward: first form vs second form (IPLAN_DEC_FWD_V2=1) on the column-grouped records
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3ac; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for v in 0 1 0 1; do
if [ $v = 1 ]; then export IPLAN_DEC_FWD_V2=1; else unset IPLAN_DEC_FWD_V2; fi
timeout 300 python scripts/microbench.py behavior_learn 2>&1 | grep -v amdgpu.ids | sed "s/^/v2=$v /" | tee -a $O/mb.txt
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> $O/bench_$v.err > $O/bench_$v.json; echo "v2=$v $(grep -o 'ms_per_step[^,]*' $O/bench_$v.json)"
done
for v in 0 1; do
if [ $v = 1 ]; then export IPLAN_DEC_FWD_V2=1; else unset IPLAN_DEC_FWD_V2; fi
( cd /tmp && IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p" -o beh -- python "$R/scripts/microbench.py" behavior_learn > /dev/null 2>&1 < /dev/null )
f=$(find $O/p -name "*kernel_stats.csv" | head -1); echo "== v2=$v"; grep "beh_dec_fwd" $f | cut -c1-120; rm -rf $O/p
done
