#!/bin/bash
# Round-2 GPU call 1: new parity tests, data-parallel single-rank check over RCCL, config-5 roofline run, kernel stats + PMC.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; O=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 > $O/rocminfo.txt 2>&1
(nproc; free -g | head -2) > $O/host.txt 2>&1
rm -f $O/parity_errors.json
timeout 1700 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 scripts/dp_single_rank_check.py > $O/dp_single_rank_check.log 2>&1; echo "rc=$?" >> $O/dp_single_rank_check.log
timeout 300 python scripts/microbench.py > $O/microbench.log 2>&1
timeout 400 python scripts/cfg5_bench.py --json $O/cfg5.json > $O/cfg5.log 2>&1; echo "rc=$?" >> $O/cfg5.log
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_cfg5" -o cfg5 -- python "$R/scripts/cfg5_bench.py" --B 256 > "$R/$O/prof_cfg5.log" 2>&1 )
python scripts/prof_summary.py $O/prof_cfg5/cfg5_kernel_stats.csv $O/cfg5_kernel_stats.csv > /dev/null 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o bench -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$R/$O/prof_bench.log" 2>&1 )
python scripts/prof_summary.py $O/prof/bench_kernel_stats.csv $O/full_cycle_kernel_stats.csv > /dev/null 2>&1
MB_PIECES="gat_fwd behavior_learn ppo_train select_actions" bash scripts/gpu_pmc.sh > $O/pmc.log 2>&1
IPLAN_BENCH_WATCHDOG=200 timeout 500 python bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
rm -rf $O/prof/*/ $O/prof_cfg5/*/ 2>/dev/null
find $O -name "*.csv" -size +3M -delete 2>/dev/null
du -sh $O > $O/du.txt
