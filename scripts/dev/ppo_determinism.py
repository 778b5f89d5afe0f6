"""Diagnostic: two identically seeded SyntheticLoops, three multi-stream cycles each, no data parallelism -- the arenas must end
bit-identical (no atomics anywhere on the path); repeats with the fp32 fc1 contraction for comparison.
python scripts/dev/ppo_determinism.py [tiny|mid]"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.harness import SyntheticLoop  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
if size == "tiny":
    kw = dict(use_cuda=True, max_vehicle_num=3, n_agents=2, episode_limit=8, ppo_epoch=2, pred_batch_size=3, max_history_len=2)
    E = 2
else:
    kw = dict(use_cuda=True, max_vehicle_num=9, n_agents=2, episode_limit=20, ppo_epoch=3, pred_batch_size=8)
    E = 4
args = default_args("highway", batch_size_run=E, buffer_size=E, batch_size=E, **kw)


def arenas_of(l):
    return [l.mac.actor_arena, l.mac.critic_arena, l.behavior.enc_arena, l.behavior.dec_arena, l.prediction.gat_arena, l.prediction.dec_arena]


def run(tag):
    out = []
    for rep in range(int(os.environ.get("REPS", "4"))):
        lp = SyntheticLoop(args, E, seed=300, device="cuda")
        lp.defer_decoder = not os.environ.get("NO_DEFER")
        if os.environ.get("SERIAL"):
            cur = torch.cuda.current_stream()
            lp._lstreams = (cur, cur)
        for c in range(3):
            torch.manual_seed(1000 + c)
            torch.cuda.manual_seed(1000 + c)
            np.random.seed(1000 + c)
            with contextlib.redirect_stdout(io.StringIO()):
                lp.cycle()
        lp.behavior.join_decoder()
        torch.cuda.synchronize()
        out.append([a.data.clone() for a in arenas_of(lp)])
    names = ["actor", "critic", "beh_enc", "beh_dec", "gat", "pred_dec"]
    for rep in range(1, len(out)):
        d = [float((a - b).abs().max()) for a, b in zip(out[0], out[rep])]
        print(tag, "run 0 vs run", rep, {n: v for n, v in zip(names, d)}, flush=True)


run(os.environ.get("TAG", "split"))
