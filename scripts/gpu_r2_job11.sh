#!/bin/bash
# A/B (same box): LDS-staged tail weights in the streaming actor/critic forward.  old = build/abl/lib_acpipe.so
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/abl; O=gpurun_out; export TMPDIR=/tmp
OLD=$R/build/abl/lib_acpipe.so
for i in 1 2; do
IPLAN_HIP_LIB=$OLD timeout 300 python scripts/microbench.py ppo_train ac_train_parts > $O/ab_old$i.log 2>&1
timeout 300 python scripts/microbench.py ppo_train ac_train_parts > $O/ab_new$i.log 2>&1
done
grep -H "gpu \|phases" $O/ab_old*.log $O/ab_new*.log > $O/abl_summary.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/abl/ppo" -o mb -- python "$R/scripts/microbench.py" ppo_train > "$R/$O/abl/ppo.log" 2>&1 )
cp $O/abl/ppo/*/mb_kernel_stats.csv $O/ppo_train_kernel_stats.csv 2>/dev/null || cp $O/abl/ppo/mb_kernel_stats.csv $O/ppo_train_kernel_stats.csv
rm -rf $O/abl/ppo
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
