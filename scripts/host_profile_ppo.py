"""Host-side cost of IPPOLearner.train at BASELINE config 3 (cProfile): which Python frames the 15 epochs spend their wall time in,
and how long the GPU idles between launches (wall time of train() against the summed kernel time of the same call).
    python scripts/host_profile_ppo.py [n_trains]"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.harness import SyntheticLoop  # noqa: E402

E = 32
args = default_args("highway", use_cuda=True, batch_size_run=E)
loop = SyntheticLoop(args, E, seed=0, device="cuda")
batch = loop.rollout()
while not loop.learner.buffers[0].can_sample():
    loop.learner.insert_episode_batch(batch)


def refill():
    while not loop.learner.buffers[0].can_sample():
        loop.learner.insert_episode_batch(batch)
    torch.cuda.synchronize()


def train():
    with contextlib.redirect_stdout(io.StringIO()):
        loop.learner.train(0)


train()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
t_issue = t_wall = 0.0
for _ in range(n):
    refill()
    t0 = time.perf_counter()
    train()
    t_issue += (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_wall += (time.perf_counter() - t0) / n
refill()
print(f"train(): host issue {t_issue * 1e3:.1f} ms, wall (incl. GPU drain) {t_wall * 1e3:.1f} ms")
pr = cProfile.Profile()
pr.enable()
train()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
