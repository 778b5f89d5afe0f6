"""Diagnostic: is Behavior_policy.learn deterministic and free of uninitialised reads?  The caching allocator's free blocks are
poisoned (NaN / huge values) before every call; the same learn() is repeated and the gradient arenas compared bit for bit.
python scripts/dev/beh_repro.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from iplan_amd import synth  # noqa: E402
from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.nova.stable_behavior_policy import Behavior_policy  # noqa: E402


class Log:
    def log_stat(self, *a, **k):
        pass


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
args = default_args("highway", use_cuda=True, batch_size_run=32)
E = 32
f = synth.make_episode_fields(args, E, 34, 0.8)
batch = synth.DictBatch(f, E, args.episode_limit + 1).to("cuda")
nA, N, Lw, T = args.n_agents, args.max_vehicle_num, args.max_history_len, args.episode_limit
J = T - 1 - Lw
gen = torch.Generator().manual_seed(35)
keep = (torch.rand(nA, J, E * N, Lw, args.decoder_rnn_dim, generator=gen) < 0.9).to(torch.uint8).cuda()
ref = None
for mode in ("nan", "big", "zero"):
    for r in range(reps):
        torch.manual_seed(33)
        pol = Behavior_policy(args, Log())
        # poison what torch.empty will hand out next
        junk = [torch.empty(int(6e9), dtype=torch.float32, device="cuda") for _ in range(8)]
        for t in junk:
            t.fill_(float("nan") if mode == "nan" else (3e30 if mode == "big" else 0.0))
        del junk
        torch.cuda.synchronize()
        pol.learn(batch, 0, keep=keep)
        pol.join_decoder()
        torch.cuda.synchronize()
        g = (pol.enc_arena.grad.clone(), pol.dec_arena.grad.clone())
        bad = [bool(torch.isnan(x).any() or torch.isinf(x).any()) for x in g]
        if ref is None:
            ref = g
        d = [float((a - b).abs().max()) for a, b in zip(g, ref)]
        rel = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(g, ref)]
        print(mode, r, "nan/inf:", bad, "max |diff| vs first run (enc, dec):", d, "rel", rel, flush=True)
        if any(x > 0 for x in d):
            diff = (g[1] - ref[1]).abs()
            for i in range(nA):
                for k in pol.dec_arena.names:
                    o = pol.dec_arena.offsets[k]
                    n = int(torch.Size(pol.dec_arena.shapes[k]).numel())
                    m = float(diff[i, o:o + n].max())
                    if m > 0:
                        print("   agent", i, k, m, "of", float(ref[1][i, o:o + n].abs().max()))
