#!/bin/bash
# bf16 wide wgrad: rows 4 j + g per K slot (256 contiguous bytes per load instruction on column-grouped rows) vs 8 g + j (previous library)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3aa; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity_fullsize.py -m gpu -x -q -k "wgrad or behavior" 2>&1 | tail -2
for rep in 1 2; do
for v in new old; do
lib=$R/iplan_amd/libiplan_hip.so; [ $v = old ] && lib=$R/build/old_tree/iplan_amd/libiplan_hip.so
IPLAN_HIP_LIB=$lib timeout 300 python scripts/microbench.py behavior_learn 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" | tee -a $O/mb.txt
IPLAN_HIP_LIB=$lib IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2> $O/bench_${v}_$rep.err > $O/bench_${v}_$rep.json; cut -c1-200 $O/bench_${v}_$rep.json
done; done
for v in new old; do
lib=$R/iplan_amd/libiplan_hip.so; [ $v = old ] && lib=$R/build/old_tree/iplan_amd/libiplan_hip.so
( cd /tmp && IPLAN_HIP_LIB=$lib IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p" -o beh -- python "$R/scripts/microbench.py" behavior_learn > /dev/null 2>&1 < /dev/null )
f=$(find $O/p -name "*kernel_stats.csv" | head -1); echo "== $v"; grep "wgrad" $f | cut -c1-140; rm -rf $O/p
done
