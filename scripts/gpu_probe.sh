#!/bin/bash
# scratch probe: stamps inside one fused vector step + rollout time (+ optional pytest subset):  bash scripts/gpu_probe.sh <tag> [pytest -k expr]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/probe; mkdir -p $O; export TMPDIR=/tmp
if [ $# -gt 1 ]; then timeout 900 python -m pytest tests -m gpu -q -x -k "$2" > $O/pytest_$1.log 2>&1; tail -3 $O/pytest_$1.log; fi
timeout 300 python scripts/dev/fused_step_clocks.py 2>&1 | grep -v amdgpu.ids > $O/clocks_$1.txt
cat $O/clocks_$1.txt
for v in base base; do timeout 200 python scripts/microbench.py rollout 2>&1 | grep rollout; done | tee $O/mb_$1.txt
