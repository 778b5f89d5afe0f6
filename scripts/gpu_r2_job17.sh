#!/bin/bash
# gat_bwd with in-kernel W_hh gradient + neighbour-major scratch (old = build/abl/lib_gatbwd_old.so: the round-2d kernel; its
# ops.gat_backward differs, so the old library is measured from the committed r02d numbers instead)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out; export TMPDIR=/tmp
for i in 1 2; do
timeout 200 python scripts/microbench.py gat_bwd_phases "gat_fwd(save)+bwd+wgrad S=64" "gat_fwd(save) S=64" prediction_learn behavior_learn > $O/ab_new$i.log 2>&1
done
grep -H "gpu \|phases" $O/ab_new*.log > $O/abl_summary.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python scripts/cfg5_bench.py --pieces gat >> $O/abl_summary.txt 2>&1
IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.log 2> $O/bench.err
