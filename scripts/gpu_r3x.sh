#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r3x; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
for kw in 4 1 2 8; do
IPLAN_AC_KSPLIT_WG=$kw timeout 300 python scripts/microbench.py select_actions rollout 2>&1 | grep -v amdgpu.ids | sed "s/^/kw$kw /" | tee -a $O/mb.txt
done; done
for kw in 4 1; do
IPLAN_AC_KSPLIT_WG=$kw IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_kw$kw.json 2> $O/bench_kw$kw.err; cut -c1-200 $O/bench_kw$kw.json
done
IPLAN_ROLLOUT_GRAPH=1 IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_graph.json 2> $O/bench_graph.err; cut -c1-200 $O/bench_graph.json
