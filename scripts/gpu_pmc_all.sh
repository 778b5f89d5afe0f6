#!/bin/bash
# PMC passes of the bench's pieces on the CURRENT build -> profiles-ready files under gpurun_out/<series>/ :
#   pmc_{rollout,behaviour_learn,ppo_train}.txt  and  <series>_pmc_summary.json (copy to profiles/: bench.py's only traffic source)
# usage: SERIES=r03c bash scripts/gpu_pmc_all.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; S=${SERIES:-r03x}; O=gpurun_out/$S; mkdir -p $O
bash scripts/gpu_pmc_piece.sh rollout gat_fwd select_actions rollout
MB_DEFER=1 bash scripts/gpu_pmc_piece.sh behaviour_learn behavior_learn     # deferred decoder update: the kernel set of the training cycle
bash scripts/gpu_pmc_piece.sh ppo_train ppo_train
for t in rollout behaviour_learn ppo_train; do mv gpurun_out/pmc_$t.txt $O/pmc_$t.txt; mv gpurun_out/pmc_$t.json $O/pmc_$t.json; done
python scripts/pmc_to_bench_json.py $S $O/pmc_rollout.json $O/pmc_behaviour_learn.json $O/pmc_ppo_train.json 32 && cp profiles/${S}_pmc_summary.json $O/
