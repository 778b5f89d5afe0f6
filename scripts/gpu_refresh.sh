#!/bin/bash
# Round deliverables in one gpurun call: GPU tests + smoke + bench line + kernel stats, then PMC passes.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
BENCH_ARGS="--steps 3 --warmup 1" bash scripts/gpu_check.sh
MB_PIECES="gat_fwd behavior_learn" bash scripts/gpu_pmc.sh
