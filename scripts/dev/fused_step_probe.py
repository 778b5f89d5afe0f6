"""The one-launch vector step against the two-launch form, field by field and step by step (which step, which field, how far)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from iplan_amd import ops  # noqa: E402
from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.harness import SyntheticLoop  # noqa: E402

T = int(os.environ.get("PROBE_T", "6"))
args = default_args("highway", use_cuda=True, episode_limit=T, batch_size_run=32)
E, dev = 32, "cuda"
loop = SyntheticLoop(args, E, seed=21, device=dev)
nA, N = args.n_agents, args.max_vehicle_num
gen = torch.Generator().manual_seed(5)
u = torch.rand(T + 1, nA, E, N, N - 1, 2, generator=gen).clamp_min(1e-20)
noise = (-torch.log((-torch.log(u)).clamp_min(1e-20))).to(dev)
q_all = (-torch.log(torch.rand(T, nA, E, args.n_actions, generator=gen).clamp_min(1e-20))).to(dev)
obs = loop.obs_sets[0]
runs = []
for rep, fuse in enumerate((True, False, True, False)):
    if fuse:
        os.environ.pop("IPLAN_NO_FUSE_AC", None)
    else:
        os.environ["IPLAN_NO_FUSE_AC"] = "1"
    b = loop.new_batch()
    with torch.no_grad():
        loop._rollout_body(obs, b, noise=noise, q_all=q_all)
    torch.cuda.synchronize()
    runs.append({k: b[k].clone() for k in ("attention_latent", "behavior_latent", "rnn_states_actors", "rnn_states_critics", "actions")})
print("sync error:", ops.fused_sync_error())
for name, (i, j) in (("fused#1 vs two-launch#1", (0, 1)), ("fused#2 vs two-launch#2", (2, 3)), ("fused#1 vs fused#2", (0, 2)), ("two-launch#1 vs #2", (1, 3))):
    print("==", name)
    for k in runs[0]:
        d = [(runs[i][k][:, t].double() - runs[j][k][:, t].double()).abs().max().item() for t in range(T + 1)]
        print(f"  {k:22s}", " ".join(f"{x:.1e}" for x in d))
