#!/bin/bash
# Kernel A/B: time microbench pieces under each variant library build/abl/lib_<v>.so (rocprofv3 kernel stats).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/abl; export TMPDIR=/tmp
for lib in build/abl/lib_*.so; do
  v=$(basename $lib .so)
  ( cd /tmp && IPLAN_HIP_LIB="$R/$lib" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/abl/$v" -o mb -- python "$R/scripts/microbench.py" $MB_PIECES > "$R/gpurun_out/abl/$v.log" 2>&1 )
  echo "== $v"; grep -E "${ABL_KERNELS:-beh_dec|beh_enc|wgrad}" "gpurun_out/abl/$v/mb_kernel_stats.csv" | awk -F, '{printf "%s calls %s avg_ns %s\n",$1,$2,$4}'
done
