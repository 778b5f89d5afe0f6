#!/bin/bash
# quick serial per-kernel timing of the behaviour learn (current build), optional env passthrough
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3j; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p" -o beh -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/serial.log" 2>&1 < /dev/null )
find $O/p -name "*kernel_stats.csv" -exec grep -E "beh_dec|beh_enc_fwd|beh_enc_bwd" {} \; | awk -F, '{printf "%s calls %s avg_ns %s\n",$1,$2,$4}'; grep behavior_learn $O/serial.log
rm -rf $O/p
timeout 300 python scripts/microbench.py behavior_learn 2>&1 | grep -v amdgpu
timeout 300 python -m pytest tests/test_gpu_parity_fullsize.py -m gpu -q -x -k "behavior" 2>&1 | tail -2
