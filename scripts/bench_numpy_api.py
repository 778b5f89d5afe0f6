"""PCIe-inclusive rate of the rollout trio through the reference-shaped numpy API (host arrays in, host arrays out every
vector step, as runners/ippo_parallel_runner.py:166-268 calls it) at BASELINE config 3 -- reported beside, never as,
bench.py's HBM-resident `value`.   python scripts/bench_numpy_api.py"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.harness import SyntheticLoop  # noqa: E402

E = 32
args = default_args("highway", use_cuda=True, batch_size_run=E)
loop = SyntheticLoop(args, E, seed=0, device=torch.device("cuda"))
with contextlib.redirect_stdout(io.StringIO()):
    batch = loop.rollout()
nA, N, L, d, Z, A = args.n_agents, args.max_vehicle_num, args.max_history_len, args.obs_shape_single, args.latent_dim, args.attention_dim
rng = np.random.default_rng(0)
hist_single = rng.uniform(-1, 1, (E, nA, N, d))
window = rng.uniform(-1, 1, (E, nA, N, L, d))
att = (rng.standard_normal((E, nA, N, A)) * 0.1).astype(np.float32)
lat = rng.dirichlet(np.ones(Z), (E, nA, N)).astype(np.float32)
eh = torch.zeros(E, 1, nA, N, args.encoder_rnn_dim, device="cuda")


def step(t):
    global att, lat, eh
    vals, acts, logps, ha, hc = loop.mac.select_actions_ippo(batch, t, test_mode=False)        # numpy out
    att = loop.prediction.GAT_latent_update(hist_single, att, lat)                             # float64 numpy in -> float32 numpy out
    lat, eh = loop.behavior.latent_update(window, eh, lat)
    return acts


for t in range(5):
    step(t)
torch.cuda.synchronize()
T = 60
t0 = time.perf_counter()
for t in range(T):
    step(t % args.episode_limit)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / T
print(f"numpy-API vector step (select_actions + GAT_latent_update + latent_update, host arrays both ways): {dt * 1e3:.2f} ms "
      f"-> {E / dt:.0f} env-steps/s rollout-only, vs ~0.27 ms per step device resident")
