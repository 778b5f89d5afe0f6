#!/bin/bash
# rocprofv3 kernel stats of selected microbench pieces -> gpurun_out/prof_mb/
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_mb" -o mb -- python "$R/scripts/microbench.py" $MB_PIECES > "$R/gpurun_out/prof_mb.log" 2>&1 )
python scripts/prof_summary.py gpurun_out/prof_mb/mb_kernel_stats.csv gpurun_out/prof_mb_summary.csv
