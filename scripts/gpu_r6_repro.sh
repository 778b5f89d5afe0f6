#!/bin/bash
# How often does a repeated Behavior_policy.learn differ bit-wise from the first run, with the side streams as created (0) and after the
# hardware-queue check (verify)?  (tests/test_gpu_fullsize.py::test_behavior_learn_bitwise_reproducible_from_a_cold_process failed once)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6repro; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for pr in 0 verify 0 verify 0 verify 0 verify 0 verify 0 verify 0 verify 0 verify; do
  echo "== IPLAN_QUEUE_PROBE=$pr" >> $O/repro.txt
  IPLAN_QUEUE_PROBE=$pr timeout 900 python scripts/dev/beh_repro.py 12 2>&1 | grep -v amdgpu.ids | grep "nan/inf\|agent" | cut -c1-170 >> $O/repro.txt
done
grep -c "nan/inf" $O/repro.txt; grep -v "\[0.0, 0.0\]" $O/repro.txt | head -40
