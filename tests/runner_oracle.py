"""TEST INFRASTRUCTURE: iplan_amd.runners.ParallelRunner (device-resident episode container, pinned staging, one D2H of the
actions per step) against the CPU oracle stepped in the reference runner's order
(runners/ippo_parallel_runner.py:105-281), with the stub vector env's stream replayed through a second history wrapper on the
host and the random draws (gumbel gate, action race) injected into both.  Every field of the episode is compared, actions
bit-equal; covers envs that terminate early (stored action 0, zero-masked state / obs / reward) and the all-terminated break.
Shared by the emulated (CPU) and the ``-m gpu`` tests."""
import numpy as np
import torch

from oracle import iplan_oracle as O


class _Log:
    def log_stat(self, *a, **k):
        pass


class _OneHot:                                                 # components/transforms.py:8-21, for the built-in container
    def __init__(self, out_dim):
        self.out_dim = out_dim

    def infer_output_info(self, vshape_in, dtype_in):
        return (self.out_dim,), torch.float32


def runner_args(device, E, nA, n_other, T, **kw):
    from iplan_amd.config import default_args
    a = default_args("highway", use_cuda=(torch.device(device).type == "cuda"), batch_size_run=E, n_agents=nA, n_other_vehicles=n_other,
                     max_vehicle_num=n_other + nA, episode_limit=T, n_obs_vehicles=6, device=str(device), animation_enable=False, **kw)
    a.obs_shape = a.obs_shape_single * a.n_obs_vehicles
    a.state_shape = a.obs_shape_single * a.max_vehicle_num
    return a


def _cpu_sd(m):
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def check_runner_vs_oracle(args, device, seed=0, end_steps=None, tol=1e-5, runs=1):
    """``runs`` > 1: the same runner object runs several episodes back to back (pinned staging buffers re-used across run()
    calls) and the LAST one is checked."""
    from iplan_amd import synth
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    from iplan_amd.nova.prediction_policy import Prediction_policy
    from iplan_amd.nova.stable_behavior_policy import Behavior_policy
    from iplan_amd.observation_wrapper import observersation_state_history_wrapper
    from iplan_amd.runners.ippo_parallel_runner import ParallelRunner, _dict_batch
    a = args
    E, nA, N, T, L = a.batch_size_run, a.n_agents, a.max_vehicle_num, a.episode_limit, a.max_history_len
    A, Z, M, R = a.attention_dim, a.latent_dim, a.rnn_hidden_dim, a.encoder_rnn_dim
    torch.manual_seed(seed)
    scheme = synth.make_scheme(a)
    scheme.pop("actions_onehot")
    scheme.pop("filled")
    full = dict(scheme, actions_onehot={"vshape": (a.n_actions,), "group": "agents"}, filled={"vshape": (1,), "dtype": torch.long})
    groups, pre = {"agents": nA}, {"actions": ("actions_onehot", [_OneHot(a.n_actions)])}
    mac = DcntrlMAC(full, groups, a)
    beh, pred = Behavior_policy(a, _Log()), Prediction_policy(a, _Log())
    mk_env = lambda s: synth.StubHighwayVecEnv(E, nA, a.n_obs_vehicles, a.obs_shape_single, N, T, seed=s, end_steps=end_steps,  # noqa: E731
                                               n_ids=max(N - 2, a.n_obs_vehicles))
    runner = ParallelRunner(a, mk_env(seed + 1), _Log())
    runner.setup(scheme, groups, pre, mac, beh, pred)
    runner.new_batch = lambda: _dict_batch(scheme, groups, E, T + 1, pre, device)      # (never the reference's container here)
    gen = torch.Generator().manual_seed(seed + 11)
    for _ in range(runs):
        u = torch.rand(T + 1, nA, E, N, N - 1, 2, generator=gen).clamp_min(1e-20)
        noise = -torch.log((-torch.log(u)).clamp_min(1e-20))
        q_all = -torch.log(torch.rand(T, nA, E, a.n_actions, generator=gen).clamp_min(1e-20))
        batch, _, avg_rwd, avg_len = runner.run(noise=noise.to(device), q_all=q_all.to(device))
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    got = {k: batch[k].cpu() for k in batch.data}

    # ---- the oracle, stepped like runners/ippo_parallel_runner.py:105-281 on a replay of the same env stream
    gat_on, beh_on = a.GAT_enable, a.Behavior_enable
    gat_p = [_cpu_sd(m) for m in pred.pred_GAT] if gat_on else None
    enc_p = [_cpu_sd(m) for m in beh.behavior_encoder] if beh_on else None
    act_p, cri_p = [_cpu_sd(m) for m in mac.agents], [_cpu_sd(m) for m in mac.critics]
    env = mk_env(seed + 1)
    hw = observersation_state_history_wrapper(a, nA, N, T, L)
    T1 = T + 1
    f32 = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32)  # noqa: E731
    ref = dict(history=torch.zeros(E, T1, nA, N, a.obs_shape_single), attention_latent=torch.zeros(E, T1, nA, N, A),
               behavior_latent=torch.zeros(E, T1, nA, N, Z), rnn_states_actors=torch.zeros(E, T1, nA, M),
               rnn_states_critics=torch.zeros(E, T1, nA, M), actions=torch.zeros(E, T1, nA, 1, dtype=torch.long),
               actions_onehot=torch.zeros(E, T1, nA, a.n_actions), reward=torch.zeros(E, T1, nA, 1),
               terminated=torch.zeros(E, T1, nA, 1, dtype=torch.uint8), state=torch.zeros(E, T1, a.state_shape),
               obs=torch.zeros(E, T1, nA, a.obs_shape), filled=torch.zeros(E, T1, 1, dtype=torch.long),
               avail_actions=torch.ones(E, T1, nA, a.n_actions, dtype=torch.int32))

    def masked(x, shape, alive):
        out = np.zeros(shape)
        out[alive] = np.asarray(x).reshape(shape)[alive]
        return f32(out)

    def gat_update(h_t, att_prev, lat_prev, nz):
        out = []
        for i in range(nA):
            x = torch.cat([h_t[:, i], lat_prev[:, i]], -1) if a.GAT_use_behavior else h_t[:, i]
            out.append(O.gat_forward(gat_p[i], x, att_prev[:, i].reshape(E * N, A), nz[i].reshape(-1, 2)).reshape(E, N, A))
        return torch.stack(out, 1)

    with torch.no_grad():
        state, obs = env.reset()
        hw.agent_obs_profile_init(obs)
        hw.obs_history_create(obs)
        single = hw.obs_single_history_output()
        state, obs = hw.pure_obs_state_wrapper(state, obs)
        terminated = np.zeros(E, dtype=bool)
        alive = np.arange(E)
        ref["history"][:, 0] = f32(single)
        ref["state"][:, 0] = masked(state, (E, a.state_shape), alive)
        ref["obs"][:, 0] = masked(obs, (E, nA, a.obs_shape), alive)
        if gat_on:
            ref["attention_latent"][:, 0] = gat_update(ref["history"][:, 0], ref["attention_latent"][:, 0], ref["behavior_latent"][:, 0], noise[T])
        ref["filled"][:, 0] = 1
        eh = torch.zeros(E, 1, nA, N, R)
        t_end = 0
        for t in range(T):
            last = ref["actions_onehot"][:, t - 1] if t > 0 else torch.zeros(E, nA, a.n_actions)
            x = O.build_inputs_rollout(ref["history"][:, t], ref["attention_latent"][:, t], ref["behavior_latent"][:, t], last, nA, gat_on, beh_on)
            for i in range(nA):
                logits, hn = O.actor_logits(act_p[i], x[:, i], ref["rnn_states_actors"][:, t, i], torch.ones(E, a.n_actions, dtype=torch.int32))
                act = (torch.softmax(logits, -1) / q_all[t, i]).argmax(-1)
                act[torch.as_tensor(terminated)] = 0                                 # action2env_tuple (:81-83, 177-180)
                ref["actions"][:, t, i, 0] = act
                ref["actions_onehot"][:, t, i] = torch.nn.functional.one_hot(act, a.n_actions).float()
                _, hcn = O.critic_value(cri_p[i], x[:, i], ref["rnn_states_critics"][:, t, i])
                ref["rnn_states_actors"][:, t + 1, i] = hn
                ref["rnn_states_critics"][:, t + 1, i] = hcn
            state, obs, reward, _, term_agent, _ = env.step([tuple(r) for r in ref["actions"][:, t, :, 0].numpy().astype(np.float64)])
            terminated = np.logical_or(np.asarray(term_agent).all(axis=1), terminated)
            if terminated.all():
                ref["rnn_states_actors"][:, t + 1] = 0                               # never stored by the reference (:212-214)
                ref["rnn_states_critics"][:, t + 1] = 0
                break
            hw.obs_history_create(obs)
            single = hw.obs_single_history_output()
            window = hw.obs_history_output() if beh_on else None
            state, obs = hw.pure_obs_state_wrapper(state, obs)
            ref["history"][:, t + 1] = f32(single)
            if gat_on:
                ref["attention_latent"][:, t + 1] = gat_update(ref["history"][:, t + 1], ref["attention_latent"][:, t], ref["behavior_latent"][:, t], noise[t])
            if beh_on:
                ref["behavior_latent"][:, t + 1], eh = O.latent_update(enc_p, f32(window), eh, ref["behavior_latent"][:, t], a.soft_update_coef)
            ref["reward"][:, t, :, 0] = masked(reward, (E, nA), alive)
            ref["terminated"][:, t, :, 0] = torch.as_tensor(np.asarray(term_agent)).to(torch.uint8)
            ref["state"][:, t + 1] = masked(state, (E, a.state_shape), alive)
            ref["obs"][:, t + 1] = masked(obs, (E, nA, a.obs_shape), alive)
            ref["filled"][:, t + 1] = 1
            alive = np.flatnonzero(~terminated)
            t_end = t + 1
    assert runner.t == t_end, (runner.t, t_end)
    for k in ("actions", "actions_onehot", "filled", "terminated", "avail_actions"):
        assert torch.equal(got[k].to(ref[k].dtype), ref[k]), k
    worst = {}
    for k in ("history", "state", "obs", "reward", "attention_latent", "behavior_latent", "rnn_states_actors", "rnn_states_critics"):
        e = (got[k].double() - ref[k].double()).abs().max().item() / max(1.0, ref[k].abs().max().item())
        worst[k] = e
        assert e < tol, (k, e)
    worst["steps"] = t_end
    return worst
