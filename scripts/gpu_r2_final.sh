#!/bin/bash
# Round-2 evidence run (one gpurun call): GPU tests + smoke, bench lines (weak default incl. cpu_baseline, strong N=1), rocprofv3
# kernel statistics of the bench command and of the serial behaviour learn, PMC passes per piece, BASELINE config 5, the
# device-resident runner with the host in the loop, the single-rank RCCL check.  Outputs -> gpurun_out/final/ (copied to profiles/history/r02f_*).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/final; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log < /dev/null
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py > $O/bench_line.json 2> $O/bench.err < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --scaling strong --no-cpu-baseline > $O/bench_strong_n1_line.json 2> $O/bench_strong.err < /dev/null
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_cycle" -o cyc -- python "$R/bench.py" --in-process --steps 2 --warmup 1 --no-cpu-baseline > "$R/$O/bench_under_rocprof.json" 2> "$R/$O/bench_under_rocprof.err" < /dev/null )
find $O/prof_cycle -name "*kernel_stats.csv" -exec cp {} $O/full_cycle_kernel_stats.csv \; ; rm -rf $O/prof_cycle
( cd /tmp && IPLAN_BEH_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_beh" -o beh -- python "$R/scripts/microbench.py" behavior_learn > "$R/$O/behaviour_serial.log" 2>&1 < /dev/null )
find $O/prof_beh -name "*kernel_stats.csv" -exec cp {} $O/behaviour_serial_kernel_stats.csv \; ; rm -rf $O/prof_beh
timeout 400 python scripts/microbench.py > $O/microbench.txt 2>&1 < /dev/null
timeout 200 python scripts/microbench.py gat_bwd_phases "gat_fwd(save)+bwd+wgrad S=64" "gat_fwd(save) S=64" defer_overlap >> $O/microbench.txt 2>&1 < /dev/null
bash scripts/gpu_pmc_piece.sh rollout select_actions rollout; mv gpurun_out/pmc_rollout.txt $O/pmc_rollout.txt < /dev/null
bash scripts/gpu_pmc_piece.sh behaviour_learn behavior_learn; mv gpurun_out/pmc_behaviour_learn.txt $O/pmc_behaviour_learn.txt < /dev/null
bash scripts/gpu_pmc_piece.sh ppo_train ppo_train; mv gpurun_out/pmc_ppo_train.txt $O/pmc_ppo_train.txt < /dev/null
timeout 400 python scripts/cfg5_bench.py --json $O/cfg5_timings.json > $O/cfg5_timings.txt 2>&1 < /dev/null
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_cfg5" -o c5 -- python "$R/scripts/cfg5_bench.py" --B 256 > /dev/null 2>&1 < /dev/null )
find $O/prof_cfg5 -name "*kernel_stats.csv" -exec cp {} $O/cfg5_B256_kernel_stats.csv \; ; rm -rf $O/prof_cfg5
timeout 300 python scripts/bench_runner.py > $O/runner_host_in_loop.txt 2>&1 < /dev/null
NCCL_DEBUG=INFO MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python scripts/dp_single_rank_check.py > $O/dp_single_rank_check_nccl.log 2>&1 < /dev/null
tail -c 3000 $O/dp_single_rank_check_nccl.log > $O/dp_tail.log; grep -m5 -i "nccl\|rccl" $O/dp_single_rank_check_nccl.log > $O/dp_head.log; cat $O/dp_head.log $O/dp_tail.log > $O/dp_single_rank_check_nccl.log; rm -f $O/dp_head.log $O/dp_tail.log
ls -la $O
