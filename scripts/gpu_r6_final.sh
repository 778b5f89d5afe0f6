#!/bin/bash
# Round-6 evidence run (one gpurun call) on the CURRENT build: PMC passes first (-> profiles/<series>_pmc_summary.json, the bench's
# traffic source), GPU tests + smoke, bench lines (weak default incl. cpu_baseline, strong N=1, rank-of-8 projection), rocprofv3
# kernel statistics of the bench command, cycle trace (busy / learn phase), microbench, BASELINE config 5, runner host-in-loop.
# usage: SERIES=r06a bash scripts/gpu_r6_final.sh     outputs -> gpurun_out/<series>/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; S=${SERIES:-r06a}; O=gpurun_out/$S; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
SERIES=$S bash scripts/gpu_pmc_all.sh > $O/pmc_all.log 2>&1 < /dev/null
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log < /dev/null
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --gpus 1 > $O/bench_line.json 2> $O/bench.err < /dev/null      # the driver's command: bench.py starts its own rank (torch.distributed.run)
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --gpus 1 --scaling strong --no-cpu-baseline > $O/bench_strong_n1_line.json 2> $O/bench_strong.err < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --scaling strong --emulate-rank-of 8 --no-cpu-baseline > $O/bench_strong_rank_of_8_projection.json 2> $O/bench_proj.err < /dev/null
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_cycle" -o cyc -- python "$R/bench.py" --in-process --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$R/$O/bench_under_rocprof.json" 2> "$R/$O/bench_under_rocprof.err" < /dev/null )
find $O/prof_cycle -name "*kernel_stats.csv" -exec cp {} $O/full_cycle_kernel_stats.csv \;
f=$(find $O/prof_cycle -name "*kernel_trace.csv" | head -1)
python scripts/trace_busy.py $f > $O/cycle_trace_busy.txt 2>&1; python scripts/trace_learn.py $f > $O/cycle_trace_learn_phase.txt 2>&1
rm -rf $O/prof_cycle
timeout 400 python scripts/microbench.py > $O/microbench.txt 2>&1 < /dev/null
timeout 400 python scripts/cfg5_bench.py --json $O/cfg5_timings.json > $O/cfg5_timings.txt 2>&1 < /dev/null
timeout 300 python scripts/bench_runner.py > $O/runner_host_in_loop.txt 2>&1 < /dev/null
echo "== IPLAN_HOST_HISTORY=1 (the id -> slot history as the numpy class on the host: round 4's form)" >> $O/runner_host_in_loop.txt
IPLAN_HOST_HISTORY=1 timeout 300 python scripts/bench_runner.py 2>&1 | grep -v amdgpu.ids >> $O/runner_host_in_loop.txt
timeout 200 python scripts/dev/fused_step_clocks.py > $O/fused_step_clocks.txt 2>&1 < /dev/null
for ag in 0 4; do timeout 600 python scripts/ppo_all_steps_check.py $O/ppo_all_steps_agent$ag.json $ag > $O/ppo_all_steps_agent$ag.log 2>&1; echo "rc=$?" >> $O/ppo_all_steps_agent$ag.log; done
for v in base nofuse; do
  if [ $v = nofuse ]; then export IPLAN_NO_FUSE_AC=1; else unset IPLAN_NO_FUSE_AC; fi
  echo "== $v" >> $O/rollout_fused_vs_two_launches.txt
  timeout 200 python scripts/microbench.py rollout select_actions 2>&1 | grep -v amdgpu.ids >> $O/rollout_fused_vs_two_launches.txt
  IPLAN_BENCH_WATCHDOG=300 timeout 400 python bench.py --in-process --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-260 >> $O/rollout_fused_vs_two_launches.txt
done
unset IPLAN_NO_FUSE_AC
ls -la $O; cut -c1-400 $O/bench_line.json; tail -3 $O/pytest_gpu.log
