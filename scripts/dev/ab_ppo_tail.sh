#!/bin/bash
# probe of the PPO epoch's tail launches: GPU parity tests of the learner, then the bench line's per-kernel timers
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/ab_tail; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "ppo or ippo or train" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2; do
  timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('step', round(d['ms_per_step'], 2), ' '.join('%s=%.1f' % (k['kernel'][:28], k['us_per_launch']) for k in d['roofline_others'] if 'ac_' in k['kernel'] and 'HBM' not in k['kernel']))" | tee -a $O/ab.txt
  timeout 100 python scripts/microbench.py ppo_train 2>&1 | grep ppo_train | tee -a $O/ab.txt
done
