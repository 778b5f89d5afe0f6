#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench line, rocprofv3 kernel stats.  Outputs -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 > gpurun_out/rocminfo.txt 2>&1
nproc > gpurun_out/nproc.txt
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
IPLAN_BENCH_WATCHDOG=120 timeout 500 python bench.py ${BENCH_ARGS:---steps 1 --warmup 1} > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --in-process --steps 1 --warmup 0 --no-cpu-baseline > "$R/gpurun_out/prof_bench.log" 2>&1 ); echo "prof rc=$?" >> gpurun_out/prof_bench.log
ls -R gpurun_out/prof | head -30 > gpurun_out/prof_ls.txt
