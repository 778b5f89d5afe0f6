#!/bin/bash
# A/B of IPLAN_DEFER_LATE (see Behavior_policy.learn): bench lines only, alternating, + the deferred-update parity test under the knob
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/ab_late; mkdir -p $O; export TMPDIR=/tmp
IPLAN_DEFER_LATE=1 timeout 600 python -m pytest tests -m gpu -q -x -k "deferred" > $O/pytest_late.log 2>&1; tail -2 $O/pytest_late.log
for rep in 1 2; do for v in 0 1 2; do
  IPLAN_DEFER_LATE=$v timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('late=$v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done; done
