#!/bin/bash
# same-box A/B of the decoder-forward counter fix: behaviour learn alone + the cycle, old library vs new
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r6ab; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2 3; do
for v in old new; do
  echo "== $v" >> $O/ab.txt
  IPLAN_HIP_LIB=$R/build/abl/lib_$v.so timeout 300 python scripts/microbench.py behavior_learn 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  IPLAN_HIP_LIB=$R/build/abl/lib_$v.so IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --in-process --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>> $O/ab.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('cycle ms_per_step %.2f' % d['ms_per_step'], [ (r['kernel'][:20], round(r['us_per_launch'],1)) for r in d['roofline_others'] if 'dec_fwd' in r['kernel']])
" >> $O/ab.txt
done; done
cat $O/ab.txt
