#!/bin/bash
# Round-3 call b: full per-tensor PPO gradient tables (non-asserting) + the new GPU runner tests.  Outputs -> gpurun_out/r3b/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3b; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_runner.py -m gpu -q -x > $O/pytest_runner.log 2>&1; echo "pytest rc=$?" >> $O/pytest_runner.log
timeout 900 python scripts/ppo_grad_error_table.py --top 40 --cases cfg3_2,switches,mb3x2 --json $O/ppo_grad_table_default.json > $O/ppo_grad_table_default.txt 2>&1
tail -5 $O/pytest_runner.log; cat $O/ppo_grad_table_default.txt | cut -c1-200
