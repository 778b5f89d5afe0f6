#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r4g; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for v in base thinrows base thinrows; do
  unset IPLAN_DEC_THIN_ROWS IPLAN_DEC_BWD_V1
  [ $v = thinrows ] && export IPLAN_DEC_THIN_ROWS=1
  echo "== $v" >> $O/repro.txt
  timeout 300 python scripts/dev/beh_repro.py 4 2>&1 | grep -v amdgpu | grep "nan/inf" >> $O/repro.txt
done
unset IPLAN_DEC_THIN_ROWS
cat $O/repro.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "behavior or deferred or properties or env_independence" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
