#!/bin/bash
# A/B of the GAT forward recurrence: fp32 MFMA (lib_0) vs split-bf16 on the matrix cores (lib_1); parity tests run on the default library.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/ab_gat; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_gat.py tests/test_gpu_rollout.py tests/test_gpu_parity_fullsize.py tests/test_gpu_learners.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
for v in 0 1 0 1; do
  echo "== lib_$v" >> $O/mb.txt
  IPLAN_HIP_LIB=$R/build/abl/lib_$v.so timeout 200 python scripts/microbench.py gat_fwd enc_fwd select_actions rollout prediction_learn >> $O/mb.txt 2>&1
done
tail -5 $O/pytest.log; cat $O/mb.txt
