#!/bin/bash
# Round-6 closing verification on the final host code (kernel sources unchanged since r06a): GPU suite, the driver-shaped line twice,
# the same with a live RCCL group (what an N > 1 rank has: queues probed by default), the rank-of-8 projection.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r06b; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --gpus 1 > $O/bench_line.json 2> $O/bench.err < /dev/null; echo "rc=$?" >> $O/bench.err
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_line_6steps.json 2>> $O/bench.err < /dev/null
IPLAN_BENCH_PG_EARLY=1 IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_line_rccl_alive_probed.json 2>> $O/bench.err < /dev/null
IPLAN_BENCH_PG_EARLY=1 IPLAN_QUEUE_PROBE=0 IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_line_rccl_alive_unprobed.json 2>> $O/bench.err < /dev/null
IPLAN_BENCH_WATCHDOG=600 timeout 700 python bench.py --scaling strong --emulate-rank-of 8 --no-cpu-baseline > $O/bench_strong_rank_of_8_projection.json 2> $O/bench_proj.err < /dev/null
for f in bench_line bench_line_6steps bench_line_rccl_alive_probed bench_line_rccl_alive_unprobed bench_strong_rank_of_8_projection; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1) lines=$(grep -c . $O/$f.json)"; done
tail -3 $O/pytest_gpu.log; tail -1 $O/smoke.log
