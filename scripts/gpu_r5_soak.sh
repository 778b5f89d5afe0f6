#!/bin/bash
# Soak run on the CURRENT build (one gpurun call, ~3 GPU-minutes): the bench cycle 200 times in one process (142 k fused vector-step
# launches with the learners' streams beside them; the cycle raises if ONE action selection ever gives up waiting for its launch's
# scenes -- ops.check_fused_sync, the forward-progress assumption of INTEGRATION.md) and the device-resident runner over 100 episodes
# with the stub simulator in the loop (9 000 vector steps, the give-up flag read at every step).  outputs -> gpurun_out/<series>/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; S=${SERIES:-r05_soak}; O=gpurun_out/$S; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
IPLAN_BENCH_WATCHDOG=900 timeout 900 python bench.py --steps 200 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_200_cycles.json 2> $O/bench_200.err < /dev/null
echo "bench rc=$?" >> $O/bench_200.err
timeout 300 python scripts/bench_runner.py --episodes 100 2>&1 | grep -v amdgpu.ids > $O/runner_100_episodes.txt
echo "runner rc=$?" >> $O/runner_100_episodes.txt
cut -c1-330 $O/bench_200_cycles.json; tail -2 $O/bench_200.err; cat $O/runner_100_episodes.txt | cut -c1-300
