#!/bin/bash
# per-kernel times of the serial behaviour learn under each build/abl/lib_*.so (rocprofv3 kernel stats)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/ab_serial; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for lib in $(ls build/abl/lib_*.so | sort -V); do
  v=$(basename $lib .so)
  ( cd /tmp && IPLAN_HIP_LIB=$R/$lib IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/p_$v" -o beh -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/$v.log" 2>&1 < /dev/null )
  echo "== $v"; find $O/p_$v -name "*kernel_stats.csv" -exec grep -E "beh_dec|beh_enc|wgrad_partial_kernel<12" {} \; | awk -F, '{printf "%s %s avg_ns %s\n",$1,$2,$4}'; grep behavior_learn $O/$v.log
  rm -rf $O/p_$v
done
