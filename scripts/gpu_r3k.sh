#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3k; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for v in v2 v1; do
  if [ $v = v1 ]; then export IPLAN_DEC_FWD_V1=1; else unset IPLAN_DEC_FWD_V1; fi
  ( cd /tmp && IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p_$v" -o beh -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/$v.log" 2>&1 < /dev/null )
  echo "== $v serial"; find $O/p_$v -name "*kernel_stats.csv" -exec grep -E "beh_dec_fwd" {} \; | awk -F, '{printf "%s calls %s avg_ns %s\n",$1,$2,$4}'
  rm -rf $O/p_$v
  timeout 300 python scripts/microbench.py behavior_learn 2>&1 | grep -v amdgpu
done
