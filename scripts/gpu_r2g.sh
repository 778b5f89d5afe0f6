#!/bin/bash
# Round-2 series g evidence (one gpurun call, after the split-bf16 GAT recurrence): GPU tests + smoke, bench lines (weak default incl.
# cpu_baseline, strong N=1), rocprofv3 kernel statistics of the bench command, microbench, PMC passes of the rollout pieces, config 5.
# Outputs -> gpurun_out/$SERIES/ (copied to profiles/r02<series>_*).  The behaviour / PPO kernels are those of series f (profiles/r02f_*).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/${SERIES:-g}; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log < /dev/null
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py > $O/bench_line.json 2> $O/bench.err < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --scaling strong --no-cpu-baseline > $O/bench_strong_n1_line.json 2> $O/bench_strong.err < /dev/null
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_cycle" -o cyc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/$O/bench_under_rocprof.json" 2> "$R/$O/bench_under_rocprof.err" < /dev/null )
find $O/prof_cycle -name "*kernel_stats.csv" -exec cp {} $O/full_cycle_kernel_stats.csv \; ; rm -rf $O/prof_cycle
timeout 400 python scripts/microbench.py > $O/microbench.txt 2>&1 < /dev/null
bash scripts/gpu_pmc_piece.sh rollout gat_fwd select_actions rollout; mv gpurun_out/pmc_rollout.txt $O/pmc_rollout.txt < /dev/null
timeout 400 python scripts/cfg5_bench.py --json $O/cfg5_timings.json > $O/cfg5_timings.txt 2>&1 < /dev/null
ls -la $O; cat $O/bench_line.json | cut -c1-600; tail -3 $O/pytest_gpu.log
