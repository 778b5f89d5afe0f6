#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/abl; O=gpurun_out; export TMPDIR=/tmp
run() { # tag lib
  ( cd /tmp && IPLAN_BEH_SERIAL=1 IPLAN_HIP_LIB=$2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/abl/$1" -o mb -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/abl/$1.log" 2>&1 )
  echo "== $1"; grep -E "beh_dec|beh_enc|wgrad_partial" "$O/abl/$1/mb_kernel_stats.csv" | awk -F, '{printf "%s calls %s avg_us %.1f\n",$1,$2,$4/1000}'
}
run base $R/iplan_amd/libiplan_hip.so > $O/abl_summary.txt
for v in 1 2 3 4 5; do run abl$v $R/build/abl/lib_abl$v.so >> $O/abl_summary.txt; done
rm -rf $O/abl/*/
