#!/bin/bash
# Round-5 quick check (one gpurun call) of the CURRENT build: the -m gpu suite (with per-test durations), smoke, the default bench line
# and the microbench pieces.     usage: SERIES=r5a bash scripts/gpu_r5_check.sh      outputs -> gpurun_out/<series>/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; S=${SERIES:-r5a}; O=gpurun_out/$S; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
nproc > $O/host.txt; free -g >> $O/host.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=15 ${PYTEST_EXTRA:-} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log < /dev/null
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err < /dev/null
timeout 400 python scripts/microbench.py > $O/microbench.txt 2>&1 < /dev/null
tail -25 $O/pytest_gpu.log; cut -c1-600 $O/bench_line.json; cat $O/microbench.txt | tail -30
