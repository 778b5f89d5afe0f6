#!/bin/bash
# Round-2 GPU call 3: new tests, phase clocks, isolated kernel timings, clean per-piece PMC, runner profile, bench (weak + strong N=1)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; O=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
rm -f $O/parity_errors.json
timeout 900 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python scripts/microbench.py gat_fwd enc_fwd select_actions rollout behavior_learn prediction_learn ppo_train gat_phases ac_phases > $O/microbench.log 2>&1
( cd /tmp && IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_serial" -o mb -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/prof_serial.log" 2>&1 )
python scripts/prof_summary.py $O/prof_serial/mb_kernel_stats.csv $O/behaviour_serial_kernel_stats.csv > /dev/null 2>&1
bash scripts/gpu_pmc_piece.sh behaviour behavior_learn > $O/pmc_b.log 2>&1
bash scripts/gpu_pmc_piece.sh ppo ppo_train > $O/pmc_p.log 2>&1
bash scripts/gpu_pmc_piece.sh rollout gat_fwd enc_fwd select_actions > $O/pmc_r.log 2>&1
timeout 200 python -m cProfile -s cumtime scripts/bench_runner.py --episodes 2 > $O/bench_runner_profile.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o bench -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$R/$O/prof_bench.log" 2>&1 )
python scripts/prof_summary.py $O/prof/bench_kernel_stats.csv $O/full_cycle_kernel_stats.csv > /dev/null 2>&1
IPLAN_BENCH_WATCHDOG=300 timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
IPLAN_BENCH_WATCHDOG=300 timeout 600 python bench.py --scaling strong --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_strong.log 2> $O/bench_strong.err; echo "bench rc=$?" >> $O/bench_strong.err
rm -rf $O/prof/*/ $O/prof_serial/*/ 2>/dev/null
find $O -name "*trace*.csv" -size +2M -delete 2>/dev/null
