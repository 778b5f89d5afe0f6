"""CPU, world_size 2, gloo: the data-parallel exchange step (host-emulated kernels).

Two things are checked on both ranks:
* EXACTNESS -- a 2-rank step equals the step of ONE process that holds the union of the ranks' data: every learner
  (behaviour, prediction, PPO) is run once on the full 4-env batch in a single-process replica and once data-parallel on
  the rank's 2-env half; the post-step parameters must agree (global loss normalisers + summed gradients);
* the replicas stay bit-identical through a full synthetic training cycle."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, io, contextlib
sys.path.insert(0, os.environ["IPLAN_ROOT"])
import numpy as np
import torch
import torch.distributed as dist
from iplan_amd import _lib as L
from tests.emu.emu_lib import get_emu_lib
L.use_library_for_tests(get_emu_lib())
from iplan_amd import synth
from iplan_amd.config import default_args
from iplan_amd.harness import SyntheticLoop
from iplan_amd.parallel import DataParallel
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
EF, ER = 4, 2                                                       # envs of the union / of one rank
kw = dict(use_cuda=False, max_vehicle_num=3, n_agents=2, episode_limit=8, ppo_epoch=2, pred_batch_size=3, max_history_len=2)
args_f = default_args("highway", batch_size_run=EF, buffer_size=EF, batch_size=EF, **kw)
args_r = default_args("highway", batch_size_run=ER, buffer_size=ER, batch_size=ER, **kw)
nA, N, T, Lw, P = args_f.n_agents, args_f.max_vehicle_num, args_f.episode_limit, args_f.max_history_len, args_f.pred_length
full = SyntheticLoop(args_f, EF, seed=100, device="cpu")            # the single-process replica (same on every rank)
loop = SyntheticLoop(args_r, ER, seed=100 + rank, device="cpu")     # this rank: different data AND initial weights ...
def arenas_of(l):
    return [l.mac.actor_arena, l.mac.critic_arena, l.behavior.enc_arena, l.behavior.dec_arena,
            l.prediction.gat_arena, l.prediction.dec_arena]
if rank == 0:
    for a, b in zip(arenas_of(loop), arenas_of(full)):
        a.data.copy_(b.data)
dp = DataParallel().attach(loop)                                    # ... until rank 0's weights are broadcast
def gathered(t):
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t.contiguous())
    return out
for a, b in zip(arenas_of(loop), arenas_of(full)):
    g = gathered(a.data)
    assert torch.equal(g[0], g[1]) and torch.equal(a.data, b.data)

# ---------------------------------------------------------------- exactness: union batch vs. 2 ranks x half
fields = synth.make_episode_fields(args_f, EF, seed=5, terminated_p=0.5)
T1 = T + 1
b_full = synth.DictBatch(fields, EF, T1)
b_rank = synth.DictBatch({k: v[rank * ER:(rank + 1) * ER].contiguous() for k, v in fields.items()}, ER, T1)
gen = torch.Generator().manual_seed(11)
def close(a, b, what):
    err = (a - b).abs().max().item() / max(1e-12, b.abs().max().item())
    assert err < 2e-6, (what, err)
# behaviour: same dropout flags on the same (env, entity) chains
J = T - 1 - Lw
keep_f = (torch.rand(nA, J, EF * N, Lw, 64, generator=gen) < 0.9).to(torch.uint8)
keep_r = keep_f[:, :, rank * ER * N:(rank + 1) * ER * N].contiguous()
lf = full.behavior.learn(b_full, 0, keep=keep_f)
lr_ = loop.behavior.learn(b_rank, 0, keep=keep_r)
close(loop.behavior.enc_arena.data, full.behavior.enc_arena.data, "behaviour encoder")
close(loop.behavior.dec_arena.data, full.behavior.dec_arena.data, "behaviour decoder")
# prediction: S samples per rank; the single process sees rank 0's samples followed by rank 1's
S, avail = args_f.pred_batch_size, T - P - 1
ep = torch.stack([torch.randint(r * ER, (r + 1) * ER, (nA, S), generator=gen) for r in range(world)], 1)   # [nA, world, S]
tt = torch.randint(0, avail, (nA, world, S), generator=gen)
sel_f = (ep * avail + tt).reshape(nA, world * S)
sel_r = ((ep[:, rank] - rank * ER) * avail + tt[:, rank])
u = torch.rand(nA, world, S, N, N - 1, 2, generator=gen).clamp_(1e-10, 1.0)
noise = -torch.log(-torch.log(u))
keep_p = (torch.rand(nA, P, world, S * N, args_f.attention_dim, generator=gen) < 0.9).float()
full.prediction.learn(b_full, 0, noise=noise.reshape(nA, world * S, N, N - 1, 2), keep=keep_p.reshape(nA, P, world * S * N, -1),
                      sel=sel_f.numpy())
loop.prediction.learn(b_rank, 0, noise=noise[:, rank].contiguous(), keep=keep_p[:, :, rank].contiguous(), sel=sel_r.numpy())
close(loop.prediction.gat_arena.data, full.prediction.gat_arena.data, "prediction GAT")
close(loop.prediction.dec_arena.data, full.prediction.dec_arena.data, "prediction decoder")
# PPO: advantage statistics, mask sums and the entropy mean run over the union's rows
with contextlib.redirect_stdout(io.StringIO()):
    full.learner.insert_episode_batch(b_full)
    full.learner.train(0)
    loop.learner.insert_episode_batch(b_rank)
    loop.learner.train(0)
close(loop.mac.actor_arena.data, full.mac.actor_arena.data, "PPO actors")
close(loop.mac.critic_arena.data, full.mac.critic_arena.data, "PPO critics")

# ---------------------------------------------------------------- config-4 trigger (bench.py --scaling strong): ONE global buffer
# of EF episodes sharded over the ranks, train() uses the FIRST EF - 1 of them (batch_size = buffer_size - 1,
# learners/ippo_learner.py:370-372): the last rank drops its last episode, so the ranks hold different row counts
from iplan_amd.learners.ippo_learner import IPPOLearner
for a, b in zip(arenas_of(loop), arenas_of(full)):
    a.data.copy_(b.data)                                            # remove the 1e-6 drift of the steps above
args_fs = default_args("highway", batch_size_run=EF, buffer_size=EF, batch_size=EF - 1, **kw)
args_rs = default_args("highway", batch_size_run=ER, buffer_size=ER, batch_size=ER - (1 if rank == world - 1 else 0), **kw)
lf_s = IPPOLearner(full.mac, full.scheme, full.logger, args_fs)
lr_s = IPPOLearner(loop.mac, loop.scheme, loop.logger, args_rs)
lr_s.dp = dp
lr_s.dp_global_rows, lr_s.dp_global_count = (EF - 1) * T, EF * T
with contextlib.redirect_stdout(io.StringIO()):
    lf_s.insert_episode_batch(b_full)
    lf_s.train(0)
    lr_s.insert_episode_batch(b_rank)
    lr_s.train(0)
close(loop.mac.actor_arena.data, full.mac.actor_arena.data, "PPO actors (strong-mode trigger)")
close(loop.mac.critic_arena.data, full.mac.critic_arena.data, "PPO critics (strong-mode trigger)")
for a in (loop.mac.actor_arena, loop.mac.critic_arena):
    g = gathered(a.data)
    assert torch.equal(g[0], g[1]), "replicas diverged (strong-mode trigger)"

# ---------------------------------------------------------------- a full synthetic cycle keeps the replicas identical
calls = []
orig = dp.all_reduce_grads
def spy(*ar):
    before = [gathered(a.grad) for a in ar]
    orig(*ar)
    for a, b in zip(ar, before):
        assert torch.allclose(a.grad, b[0] + b[1], rtol=0, atol=1e-7)
    calls.append(len(ar))
dp.all_reduce_grads = spy
with contextlib.redirect_stdout(io.StringIO()):
    loop.cycle()
assert len(calls) == 1 + 1 + args_r.ppo_epoch, calls                # behaviour, prediction, one per PPO epoch
for a in arenas_of(loop):
    g = gathered(a.data)
    assert torch.equal(g[0], g[1]), "replicas diverged"
    assert torch.isfinite(a.data).all()
# ---------------------------------------------------------------- the deferred decoder update (what the GPU loop runs): encoder
# and decoder arenas are all-reduced in two calls, the decoder's behind the encoder's optimiser step
calls.clear()
loop.defer_decoder = True
with contextlib.redirect_stdout(io.StringIO()):
    loop.cycle()
loop.behavior.join_decoder()
assert calls == [1, 1, 2] + [2] * args_r.ppo_epoch, calls     # behaviour: encoder, then decoder; prediction; one per PPO epoch
for a in arenas_of(loop):
    g = gathered(a.data)
    assert torch.equal(g[0], g[1]), "replicas diverged (deferred decoder update)"
    assert torch.isfinite(a.data).all()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_data_parallel_two_ranks_gloo(tmp_path):
    from tests.emu.emu_lib import get_emu_lib
    get_emu_lib()                                               # build the emulated library once, before the ranks race for it
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, IPLAN_ROOT=ROOT, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2
