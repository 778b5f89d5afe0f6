#!/bin/bash
# A/B of the behaviour decoder kernels (build/abl/lib_<v>.so: IPLAN_DEC_BF3 = 0 fp32 MFMA, 1 split-bf16) + parity tests on the default library.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/ab_dec; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_learners.py tests/test_gpu_parity_fullsize.py tests/test_gpu_fullsize.py tests/test_gpu_rollout.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
for v in 0 1 0 1; do
  echo "== lib_$v" >> $O/mb.txt
  IPLAN_HIP_LIB=$R/build/abl/lib_$v.so timeout 200 python scripts/microbench.py behavior_learn rollout 2>&1 | grep -v amdgpu.ids >> $O/mb.txt
done
( cd /tmp && IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_beh" -o beh -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/behaviour_serial.log" 2>&1 < /dev/null )
find $O/prof_beh -name "*kernel_stats.csv" -exec cp {} $O/behaviour_serial_kernel_stats.csv \; ; rm -rf $O/prof_beh
tail -5 $O/pytest.log; cat $O/mb.txt; head -12 $O/behaviour_serial_kernel_stats.csv | cut -c1-150
