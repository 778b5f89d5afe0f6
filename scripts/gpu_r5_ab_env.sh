#!/bin/bash
# Same-box A/B of ENVIRONMENT knobs on microbench pieces and the short training cycle (one gpurun call):
#   VARIANTS="base:; nopair:IPLAN_WG_NO_PAIR=1" MB_PIECES="behavior_learn ppo_train" REPS=2 CYCLE=1 bash scripts/gpu_r5_ab_env.sh
# every variant = name:ENV=VAL,ENV=VAL ; MB_DEFER=1 is set for the microbench (the training cycle's form of Behavior_policy.learn)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/${SERIES:-abenv}; mkdir -p $O; export TMPDIR=/tmp
: > $O/ab.txt
IFS=';' read -ra VS <<< "${VARIANTS:-base:}"
for rep in $(seq 1 ${REPS:-2}); do
for v in "${VS[@]}"; do
  v=$(echo "$v" | xargs); name=${v%%:*}; envs=${v#*:}
  echo "== $name   [$envs]" >> $O/ab.txt
  ( IFS=','; for e in $envs; do [ -n "$e" ] && export "$e"; done; unset IFS
    MB_DEFER=1 timeout 300 python scripts/microbench.py ${MB_PIECES:-behavior_learn} 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
    if [ -n "${CYCLE:-}" ]; then IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cycle ms_per_step', d['ms_per_step'], 'value', d['value'])" >> $O/ab.txt; fi )
done; done
cat $O/ab.txt
