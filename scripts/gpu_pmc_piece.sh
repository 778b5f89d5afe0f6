#!/bin/bash
# PMC counters of ONE microbench piece set (separate passes; kernel-trace only beside --pmc): scripts/gpu_pmc_piece.sh <tag> <pieces...>
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; tag=$1; shift
mkdir -p gpurun_out/pmc_$tag; export TMPDIR=/tmp
i=0
while read -r line; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $line --output-format csv -d "$R/gpurun_out/pmc_$tag/p$i" -o pmc -- python "$R/scripts/microbench.py" "$@" > "$R/gpurun_out/pmc_$tag/p$i.log" 2>&1 )
done <<PASSES
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE
PASSES
python scripts/pmc_summary.py gpurun_out/pmc_$tag --json gpurun_out/pmc_$tag.json > gpurun_out/pmc_$tag.txt 2>&1
rm -rf gpurun_out/pmc_$tag
