"""TEST INFRASTRUCTURE: one rank of the run_ippo-side data-parallel recipe of INTEGRATION.md section 4, launched by
``python -m torch.distributed.run --nproc-per-node 2`` from tests/test_dp_runner_gloo.py (CPU, host-emulated kernels, gloo).

Every rank builds what ``run_ippo.run_sequential`` builds (run_ippo.py:194-222) -- ``DcntrlMAC``, ``IPPOLearner``,
``Behavior_policy``, ``Prediction_policy`` -- plus the device-resident ``ParallelRunner`` around ITS shard of the parallel
environments (a stub vector env: ``StubHighwayVecEnv.shard``), from the namespace ``parallel.shard_args`` derives for the
rank, and hooks them with ``DataParallel.attach(mac=, learner=, behavior=, prediction=, runner=)``.  Then one iteration of
the reference loop (run_ippo.py:261-286): ``runner.run`` -> ``insert_episode_batch`` -> ``Behavior_policy.learn`` ->
``Prediction_policy.learn`` -> ``IPPOLearner.train``.

Checked on both ranks against ONE process that runs the same loop on the union (4 envs, the whole vector env, no hooks):
* the rank's episode batch is the union's batch restricted to the rank's envs (the rollout needs no communication);
* ``runner.t_env`` and the logged episode averages are the union's on every rank, although the ranks' envs terminate at
  different steps -- and the linear lr decay driven by ``t_env`` (learners/ippo_learner.py:86-91) therefore takes the
  union's value;
* after the three learners the parameters equal the union's (global loss normalisers + summed gradients; the strong-mode
  trigger: a global buffer of which the LAST rank drops its last episode) and the replicas are bit-identical.
"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.environ["IPLAN_ROOT"])
import numpy as np                                             # noqa: E402
import torch                                                   # noqa: E402
import torch.distributed as dist                               # noqa: E402

from iplan_amd import _lib as L                                # noqa: E402
from tests.emu.emu_lib import get_emu_lib                      # noqa: E402

L.use_library_for_tests(get_emu_lib())
from iplan_amd import parallel, synth                          # noqa: E402
from iplan_amd.controllers.dcntrl_controller import DcntrlMAC  # noqa: E402
from iplan_amd.learners.ippo_learner import IPPOLearner        # noqa: E402
from iplan_amd.nova.prediction_policy import Prediction_policy  # noqa: E402
from iplan_amd.nova.stable_behavior_policy import Behavior_policy  # noqa: E402
from iplan_amd.runners.ippo_parallel_runner import ParallelRunner, _dict_batch  # noqa: E402
from tests.runner_oracle import _OneHot, runner_args            # noqa: E402


class Log:
    def __init__(self):
        self.stats = {}

    def log_stat(self, k, v, t):
        self.stats[k] = (v, t)


def build(args, env, seed):
    """what run_ippo.run_sequential builds, in its order (run_ippo.py:141-222)"""
    torch.manual_seed(seed)
    logger = Log()
    runner = ParallelRunner(args, env, logger)
    scheme = synth.make_scheme(args)
    scheme.pop("actions_onehot")
    scheme.pop("filled")
    full = dict(scheme, actions_onehot={"vshape": (args.n_actions,), "group": "agents"}, filled={"vshape": (1,), "dtype": torch.long})
    groups, pre = {"agents": args.n_agents}, {"actions": ("actions_onehot", [_OneHot(args.n_actions)])}
    mac = DcntrlMAC(full, groups, args)
    learner = IPPOLearner(mac, full, logger, args)
    behavior = Behavior_policy(args, logger)
    prediction = Prediction_policy(args, logger)
    runner.setup(scheme=scheme, groups=groups, preprocess=pre, mac=mac, behavior_learner=behavior, prediction_learner=prediction)
    E, T = args.batch_size_run, args.episode_limit
    runner.new_batch = lambda: _dict_batch(scheme, groups, E, T + 1, pre, "cpu")          # (never the reference's container here)
    return dict(runner=runner, mac=mac, learner=learner, behavior=behavior, prediction=prediction, logger=logger)


def arenas_of(o):
    return [o["mac"].actor_arena, o["mac"].critic_arena, o["behavior"].enc_arena, o["behavior"].dec_arena,
            o["prediction"].gat_arena, o["prediction"].dec_arena]


dp = parallel.init_from_env(device="cpu")                      # gloo on a CPU device; "nccl" (= RCCL) on cuda:LOCAL_RANK
rank, world = dp.rank, dp.world
assert world == 2 and dist.get_backend() == "gloo"
EF = 4
ER = EF // world
nA, n_other, T = 2, 5, 8
union_args = runner_args("cpu", EF, nA, n_other, T, max_history_len=2, pred_batch_size=3, ppo_epoch=2,
                         buffer_size=EF, batch_size=EF - 1,                      # strong mode: ONE global buffer, its last episode unused
                         use_linear_lr_decay=True, t_max=200)                    # lr depends on t_env: it must be the union's on every rank
rank_args = parallel.shard_args(union_args, world, rank, "strong")
assert (rank_args.batch_size_run, rank_args.buffer_size) == (ER, ER)
assert rank_args.batch_size == ER - (1 if rank == world - 1 else 0)
N, Lw, P = union_args.max_vehicle_num, union_args.max_history_len, union_args.pred_length
end_steps = [99, 99, 99, 4]                                     # env 3 (rank 1's) terminates early: the ranks count different env steps
union_env = synth.StubHighwayVecEnv(EF, nA, union_args.n_obs_vehicles, union_args.obs_shape_single, N, T, seed=7, end_steps=end_steps,
                                    n_ids=max(N - 2, union_args.n_obs_vehicles))
full = build(union_args, union_env, seed=100)                   # the single-process union (the same on every rank)
mine = build(rank_args, union_env.shard(rank * ER, (rank + 1) * ER), seed=100 + rank)    # this rank: its own initial weights ...
if rank == 0:
    for a, b in zip(arenas_of(mine), arenas_of(full)):
        a.data.copy_(b.data)
dp.attach(mac=mine["mac"], learner=mine["learner"], behavior=mine["behavior"], prediction=mine["prediction"], runner=mine["runner"])
for a, b in zip(arenas_of(mine), arenas_of(full)):              # ... until attach() broadcast rank 0's
    assert torch.equal(a.data, b.data)
assert mine["learner"].dp_global_rows == (EF - 1) * T and mine["learner"].dp_global_count == EF * T


def gathered(t):
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t.contiguous())
    return out


def close(a, b, what, tol=2e-6):
    err = (a - b).abs().max().item() / max(1e-12, b.abs().max().item())
    assert err < tol, (what, err)


# ------------------------------------------------------------------ runner.run(): the rank's shard of the union's rollout
gen = torch.Generator().manual_seed(11)
u = torch.rand(T + 1, nA, EF, N, N - 1, 2, generator=gen).clamp_min(1e-20)
noise = -torch.log((-torch.log(u)).clamp_min(1e-20))
q_all = -torch.log(torch.rand(T, nA, EF, union_args.n_actions, generator=gen).clamp_min(1e-20))
sl = slice(rank * ER, (rank + 1) * ER)
b_full, wf, rf, lf = full["runner"].run(noise=noise, q_all=q_all)
b_mine, wm, rm, lm = mine["runner"].run(noise=noise[:, :, sl].contiguous(), q_all=q_all[:, :, sl].contiguous())
for k in b_full.data:
    want, got = b_full[k][sl], b_mine[k]
    if want.dtype.is_floating_point:
        close(got, want, f"episode field {k}", tol=1e-6)
    else:
        assert torch.equal(got, want), k
steps = gathered(torch.tensor([mine["runner"].env_steps_this_run]))
assert int(steps[0]) != int(steps[1]), "the test wants ranks with different local step counts"
assert mine["runner"].t_env == full["runner"].t_env == int(steps[0]) + int(steps[1]), (mine["runner"].t_env, full["runner"].t_env, steps)
assert abs(rm - rf) < 1e-9 and abs(lm - lf) < 1e-9 and abs(wm - wf) < 1e-9, "logged episode averages are the union's"
t_env = mine["runner"].t_env

# ------------------------------------------------------------------ the three learners (run_ippo.py:266-286), same injected draws
J = T - 1 - Lw
keep_f = (torch.rand(nA, J, EF * N, Lw, 64, generator=gen) < 0.9).to(torch.uint8)
keep_r = keep_f[:, :, rank * ER * N:(rank + 1) * ER * N].contiguous()
S, avail = union_args.pred_batch_size, T - P - 1
ep = torch.stack([torch.randint(r * ER, (r + 1) * ER, (nA, S), generator=gen) for r in range(world)], 1)       # [nA, world, S]
tt = torch.randint(0, avail, (nA, world, S), generator=gen)
sel_f = (ep * avail + tt).reshape(nA, world * S)
sel_r = (ep[:, rank] - rank * ER) * avail + tt[:, rank]
un = torch.rand(nA, world, S, N, N - 1, 2, generator=gen).clamp_(1e-10, 1.0)
pn = -torch.log(-torch.log(un))
keep_p = (torch.rand(nA, P, world, S * N, union_args.attention_dim, generator=gen) < 0.9).float()
with contextlib.redirect_stdout(io.StringIO()):
    for o, b, kb, kw in ((full, b_full, keep_f, dict(noise=pn.reshape(nA, world * S, N, N - 1, 2), keep=keep_p.reshape(nA, P, world * S * N, -1),
                                                      sel=sel_f.numpy())),
                         (mine, b_mine, keep_r, dict(noise=pn[:, rank].contiguous(), keep=keep_p[:, :, rank].contiguous(), sel=sel_r.numpy()))):
        o["learner"].insert_episode_batch(b)
        o["behavior"].learn(b, t_env, keep=kb)
        o["prediction"].learn(b, t_env, **kw)
        o["learner"].train(t_env)
names = ("PPO actors", "PPO critics", "behaviour encoder", "behaviour decoder", "prediction GAT", "prediction decoder")
for a, b, what in zip(arenas_of(mine), arenas_of(full), names):
    close(a.data, b.data, what)
    g = gathered(a.data)
    assert torch.equal(g[0], g[1]), ("replicas diverged", what)
    assert torch.isfinite(a.data).all()
lr_f, lr_m = (o["learner"].actor_optimizers[0].param_groups[0]["lr"] for o in (full, mine))
assert lr_f == lr_m and lr_m < union_args.lr, ("the lr decay took the union's t_env", lr_f, lr_m)
assert mine["learner"].store.count == 0 and full["learner"].store.count == 0, "train() ran (and cleared the buffer) on both"
dist.destroy_process_group()
print("rank", rank, "ok")
