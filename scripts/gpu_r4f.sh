#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r4f; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for v in base thinrows bwdv1; do
  unset IPLAN_DEC_THIN_ROWS IPLAN_DEC_BWD_V1
  [ $v = thinrows ] && export IPLAN_DEC_THIN_ROWS=1
  [ $v = bwdv1 ] && export IPLAN_DEC_BWD_V1=1
  echo "== $v" >> $O/repro.txt
  timeout 300 python scripts/dev/beh_repro.py 3 2>&1 | grep -v amdgpu >> $O/repro.txt
done
cat $O/repro.txt
