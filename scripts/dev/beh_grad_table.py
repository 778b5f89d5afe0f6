"""Diagnostic: per-tensor clipped-gradient errors of Behavior_policy.learn at config 3 (E = 32, whole episode) for chosen agents
vs the fp64 oracle, under the current environment knobs.  python scripts/dev/beh_grad_table.py [agents ...] [--seed S]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from iplan_amd.config import default_args  # noqa: E402
from tests.oracle_checks import check_behavior_learn_vs_oracle  # noqa: E402

argv = [a for a in sys.argv[1:]]
seed = 33
if "--seed" in argv:
    seed = int(argv[argv.index("--seed") + 1])
    del argv[argv.index("--seed"):argv.index("--seed") + 2]
agents = tuple(int(a) for a in argv) or (2,)
torch.set_num_threads(min(16, os.cpu_count() or 1))
table = []
w = check_behavior_learn_vs_oracle(default_args("highway", use_cuda=True, batch_size_run=32), 32, "cuda", seed=seed, agents=agents, table=table)
print({k: os.environ.get(k) for k in ("IPLAN_DEC_BWD_V1", "IPLAN_DEC_THIN_ROWS", "IPLAN_ENC_FP32", "IPLAN_DEC_FWD_V1")}, "loss err", w["loss"])
for r in table:
    print(f"agent {r['agent']} {r['net']:3s} {r['tensor']:28s} kernel {r['kernel']:.2e}  fp32 oracle {r['fp32_oracle']:.2e}  max|g| {r['gmax']:.3e}")
