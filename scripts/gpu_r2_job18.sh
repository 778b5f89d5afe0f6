#!/bin/bash
# encoder forward: x fetched one step ahead (old = build/abl/lib_encfwd_old.so)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out; export TMPDIR=/tmp
OLD=$R/build/abl/lib_encfwd_old.so
run() { ( cd /tmp && IPLAN_BEH_SERIAL=1 IPLAN_HIP_LIB=$2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/abl/$1" -o mb -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/abl/$1.log" 2>&1 )
  echo "== $1 (serial)"; grep -E "beh_dec|beh_enc" $(find "$O/abl/$1" -name "*kernel_stats.csv") | awk -F, '{printf "%s calls %s avg_us %.1f\n",$1,$2,$4/1000}'; }
mkdir -p $O/abl
run old $OLD > $O/abl_summary.txt
run new $R/iplan_amd/libiplan_hip.so >> $O/abl_summary.txt
for i in 1 2; do
IPLAN_HIP_LIB=$OLD timeout 200 python scripts/microbench.py behavior_learn > $O/ab_old$i.log 2>&1
timeout 200 python scripts/microbench.py behavior_learn > $O/ab_new$i.log 2>&1
done
grep -H "behavior_learn" $O/ab_old*.log $O/ab_new*.log >> $O/abl_summary.txt
rm -rf $O/abl/*/
