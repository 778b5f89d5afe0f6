#!/bin/bash
# timing ablations of the split-bf16 fc1 forward kernel (build/abl/lib_abl*.so, results wrong by construction)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3r; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity_fullsize.py -m gpu -x -q -k "fc1_split" 2>&1 | tail -2
for v in base nw8rt2 nw4rt3 nw8rt4; do
lib=$R/iplan_amd/libiplan_hip.so; [ $v != base ] && lib=$R/build/abl/lib_$v.so
( cd /tmp && IPLAN_HIP_LIB=$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p" -o ppo -- python "$R/scripts/microbench.py" ppo_train > /dev/null 2> "$R/$O/prof.err" < /dev/null )
f=$(find $O/p -name "*kernel_stats.csv" | head -1); echo "== $v" | tee -a $O/abl.txt; grep "split" $f | cut -c1-120 | tee -a $O/abl.txt; rm -rf $O/p
done
