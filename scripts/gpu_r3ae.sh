#!/bin/bash
# can the next rollout's first launches get past the deferred wide wgrad?  shorter wgrad waves (5 rounds instead of 1) x a high-priority cycle stream
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3ae; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for rep in 1 2; do
for v in base prio chunks chunks_prio; do
lib=$R/iplan_amd/libiplan_hip.so; case $v in chunks*) lib=$R/build/abl/lib_chunks512.so;; esac
unset IPLAN_WORK_PRIORITY; case $v in *prio) export IPLAN_WORK_PRIORITY=1;; esac
IPLAN_HIP_LIB=$lib IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> $O/bench_${v}_$rep.err > $O/bench_${v}_$rep.json; echo "$v $(grep -o 'ms_per_step[^,]*' $O/bench_${v}_$rep.json)"
done; done
export IPLAN_WORK_PRIORITY=1
( cd /tmp && IPLAN_HIP_LIB=$R/build/abl/lib_chunks512.so timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p" -o cyc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 < /dev/null )
f=$(find $O/p -name "*kernel_trace.csv" | head -1); python scripts/trace_learn.py $f > $O/cycle_trace_learn_phase.txt; grep -n "wgrad_partial_bf16\|gat_fwd_kernel\|gumbel\|ac_fwd" $O/cycle_trace_learn_phase.txt | head -8 | cut -c1-120; python scripts/trace_busy.py $f | tail -9; rm -rf $O/p
