"""TEST INFRASTRUCTURE: the data-parallel worker of tests/test_dp_gloo.py (CPU, host-emulated kernels, gloo) and
tests/test_dp_gpu_cycle.py (two processes on ONE MI355X, the real kernels and HIP streams, gradient arenas through the
peer-to-peer all-reduce over HIP IPC mappings, control plane on gloo).  Launched by ``python -m torch.distributed.run``;
``IPLAN_DP_DEVICE`` = cpu | cuda.

Two things are checked on both ranks:
* EXACTNESS -- a 2-rank step equals the step of ONE process that holds the union of the ranks' data: every learner
  (behaviour, prediction, PPO) is run once on the full 4-env batch in a single-process replica and once data-parallel on
  the rank's 2-env half; the post-step parameters must agree (global loss normalisers + summed gradients);
* the replicas stay bit-identical through full synthetic training cycles (harness.SyntheticLoop.cycle: on the GPU that is the
  multi-stream loop bench.py times -- prediction / PPO learner streams, the deferred decoder update's side stream -- so the
  collectives are issued from four producer streams)."""
import os, sys, io, contextlib
sys.path.insert(0, os.environ["IPLAN_ROOT"])
import numpy as np
import torch
import torch.distributed as dist
DEV = os.environ.get("IPLAN_DP_DEVICE", "cpu")
from iplan_amd import _lib as L
if DEV == "cpu":
    from tests.emu.emu_lib import get_emu_lib
    L.use_library_for_tests(get_emu_lib())
else:
    torch.cuda.set_device(0)                                        # BOTH ranks on cuda:0
from iplan_amd import synth
from iplan_amd.config import default_args
from iplan_amd.harness import SyntheticLoop
from iplan_amd.parallel import DataParallel
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
EF, ER = 4, 2                                                       # envs of the union / of one rank
kw = dict(use_cuda=(DEV != "cpu"), max_vehicle_num=3, n_agents=2, episode_limit=8, ppo_epoch=2, pred_batch_size=3, max_history_len=2)
args_f = default_args("highway", batch_size_run=EF, buffer_size=EF, batch_size=EF, **kw)
args_r = default_args("highway", batch_size_run=ER, buffer_size=ER, batch_size=ER, **kw)
nA, N, T, Lw, P = args_f.n_agents, args_f.max_vehicle_num, args_f.episode_limit, args_f.max_history_len, args_f.pred_length
full = SyntheticLoop(args_f, EF, seed=100, device=DEV)            # the single-process replica (same on every rank)
loop = SyntheticLoop(args_r, ER, seed=100 + rank, device=DEV)     # this rank: different data AND initial weights ...
def arenas_of(l):
    return [l.mac.actor_arena, l.mac.critic_arena, l.behavior.enc_arena, l.behavior.dec_arena,
            l.prediction.gat_arena, l.prediction.dec_arena]
if rank == 0:
    for a, b in zip(arenas_of(loop), arenas_of(full)):
        a.data.copy_(b.data)
dp = DataParallel().attach(loop)                                    # ... until rank 0's weights are broadcast
def gathered(t):
    t = t.detach().cpu().contiguous()                               # (control plane on gloo: through the host)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return out
def dev(t):
    return t.to(DEV)
for a, b in zip(arenas_of(loop), arenas_of(full)):
    g = gathered(a.data)
    assert torch.equal(g[0], g[1]) and torch.equal(a.data, b.data)
if DEV != 'cpu':
    torch.cuda.synchronize()

# ---------------------------------------------------------------- exactness: union batch vs. 2 ranks x half
fields = synth.make_episode_fields(args_f, EF, seed=5, terminated_p=0.5)
T1 = T + 1
b_full = synth.DictBatch(fields, EF, T1).to(DEV)
b_rank = synth.DictBatch({k: v[rank * ER:(rank + 1) * ER].contiguous() for k, v in fields.items()}, ER, T1).to(DEV)
gen = torch.Generator().manual_seed(11)
def close(a, b, what):
    err = (a - b).abs().max().item() / max(1e-12, b.abs().max().item())
    assert err < 2e-6, (what, err)
# behaviour: same dropout flags on the same (env, entity) chains
J = T - 1 - Lw
keep_f = (torch.rand(nA, J, EF * N, Lw, 64, generator=gen) < 0.9).to(torch.uint8)
keep_r = keep_f[:, :, rank * ER * N:(rank + 1) * ER * N].contiguous()
lf = full.behavior.learn(b_full, 0, keep=dev(keep_f))
lr_ = loop.behavior.learn(b_rank, 0, keep=dev(keep_r))
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
full.prediction.learn(b_full, 0, noise=dev(noise.reshape(nA, world * S, N, N - 1, 2)), keep=dev(keep_p.reshape(nA, P, world * S * N, -1)),
                      sel=sel_f.numpy())
loop.prediction.learn(b_rank, 0, noise=dev(noise[:, rank].contiguous()), keep=dev(keep_p[:, :, rank].contiguous()), sel=sel_r.numpy())
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

# ---------------------------------------------------------------- num_mini_batch = 2 under data parallelism: the minibatches are the
# UNION's (rank 0 draws the reference's permutations of the global row range and broadcasts them; every rank trains on the rows
# that fall into its own range, padded to a common launch size at weight 0)
for a, b in zip(arenas_of(loop), arenas_of(full)):
    a.data.copy_(b.data)
args_fm = default_args("highway", batch_size_run=EF, buffer_size=EF, batch_size=EF, num_mini_batch=2, **kw)
args_rm = default_args("highway", batch_size_run=ER, buffer_size=ER, batch_size=ER, num_mini_batch=2, **kw)
lf_m = IPPOLearner(full.mac, full.scheme, full.logger, args_fm)
lr_m = IPPOLearner(loop.mac, loop.scheme, loop.logger, args_rm)
lr_m.dp = dp
with contextlib.redirect_stdout(io.StringIO()):
    torch.manual_seed(77)
    lf_m.insert_episode_batch(b_full)
    lf_m.train(0)
    torch.manual_seed(77)
    lr_m.insert_episode_batch(b_rank)
    lr_m.train(0)
close(loop.mac.actor_arena.data, full.mac.actor_arena.data, "PPO actors (2 minibatches)")
close(loop.mac.critic_arena.data, full.mac.critic_arena.data, "PPO critics (2 minibatches)")
for a in (loop.mac.actor_arena, loop.mac.critic_arena):
    g = gathered(a.data)
    assert torch.equal(g[0], g[1]), "replicas diverged (2 minibatches)"

# ---------------------------------------------------------------- a full synthetic cycle keeps the replicas identical
calls = []
orig = dp.all_reduce_grads
def spy(*ar):
    before = [gathered(a.grad) for a in ar]
    orig(*ar)
    for a, b in zip(ar, before):
        assert torch.allclose(a.grad.cpu(), b[0] + b[1], rtol=0, atol=1e-7)
    calls.append(len(ar))
dp.all_reduce_grads = spy
with contextlib.redirect_stdout(io.StringIO()):
    loop.cycle()
assert sum(calls) == 4 + 2 * args_r.ppo_epoch, calls               # every gradient arena once per optimiser step
if DEV == "cpu":
    assert len(calls) == 1 + 1 + args_r.ppo_epoch, calls            # behaviour, prediction, one call per PPO epoch
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
assert sum(calls) == 4 + 2 * args_r.ppo_epoch, calls
if DEV == "cpu":
    assert calls == [1, 1, 2] + [2] * args_r.ppo_epoch, calls     # behaviour: encoder, then decoder; prediction; one per PPO epoch
else:
    assert sorted(calls) == [1, 1, 2] + [2] * args_r.ppo_epoch, calls   # (the GPU loop enqueues the side learners first)
for a in arenas_of(loop):
    g = gathered(a.data)
    assert torch.equal(g[0], g[1]), "replicas diverged (deferred decoder update)"
    assert torch.isfinite(a.data).all()

# ---------------------------------------------------------------- GPU only: collective STREAM ORDERING.  Two identically seeded
# loops per rank run the multi-stream training cycle (deferred decoder update: collectives issued from four producer streams)
# -- A through the P2P all-reduce on its communication stream, B through the host (synchronous: ordered by construction) -- with
# NO host synchronisation added by the test.  Two ranks sum commutatively, so A and B must end bit-identical; a collective that
# raced its producer stream, or a staging half overwritten early, shows up as a difference.
if DEV != "cpu":
    dp.all_reduce_grads = orig
    pair = []
    for use_p2p in (True, False):
        lp = SyntheticLoop(args_r, ER, seed=300 + rank, device=DEV)
        d2 = DataParallel()
        d2.use_p2p = use_p2p
        d2.attach(lp)
        lp.defer_decoder = True
        for c in range(3):
            torch.manual_seed(1000 + c)
            torch.cuda.manual_seed(1000 + c)
            np.random.seed(1000 + c)                                # (the prediction learner's sample draw: np.random.choice)
            with contextlib.redirect_stdout(io.StringIO()):
                lp.cycle()
        lp.behavior.join_decoder()
        torch.cuda.synchronize()
        if d2.p2p is not None:
            d2.p2p.check_error()
        pair.append([a.data.clone() for a in arenas_of(lp)])
        assert (d2.p2p is not None) == use_p2p
        d2.close()
    for k, (a, b) in enumerate(zip(*pair)):
        assert torch.equal(a, b), ("P2P multi-stream cycle differs from the host-ordered one", k, float((a - b).abs().max()))
        g = gathered(a)
        assert torch.equal(g[0], g[1]), "replicas diverged (P2P cycles)"
    assert dp.p2p is not None, "the exactness checks above were meant to go through the P2P path"
    dp.p2p.check_error()
    dp.close()
dist.destroy_process_group()
print("rank", rank, "ok")
