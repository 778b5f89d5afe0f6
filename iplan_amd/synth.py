"""Synthetic episode-buffer inputs with the value distributions of SURVEY.md §8(d).

The real simulators (Heterogeneous_Highway_Env, MPE) are not installable offline, so tests,
``bench.py`` and the golden-vector generator all drive the hot path with these tensors.  The
container mirrors the subset of pymarl's ``EpisodeBatch`` interface the hot path touches
(``batch[key]``, ``batch.batch_size``, ``batch.device``, ``batch.max_t_filled()``,
components/episode_buffer.py:118-130,200-201) so the reference's own ``EpisodeBatch`` and this
stand-in are interchangeable.
"""
import numpy as np
import torch


class DictBatch:
    """Duck-typed stand-in for components.episode_buffer.EpisodeBatch (read side only)."""

    def __init__(self, data, batch_size, max_seq_length, device="cpu"):
        self.data = data
        self.batch_size = batch_size
        self.max_seq_length = max_seq_length
        self.device = device

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.data[key]
        raise ValueError("DictBatch only supports string keys")

    def max_t_filled(self):
        return int(torch.sum(self.data["filled"], 1).max().item())

    def to(self, device):
        return DictBatch({k: v.to(device) for k, v in self.data.items()}, self.batch_size,
                         self.max_seq_length, device)


def make_history(gen, shape_prefix, N, d, presence_p=0.3):
    """Per-entity kinematic rows: column 0 = presence in {0,1}; absent entities are all-zero."""
    feats = torch.rand(*shape_prefix, N, d, generator=gen) * 2 - 1
    present = (torch.rand(*shape_prefix, N, 1, generator=gen) < presence_p).float()
    feats[..., 0:1] = 1.0
    return feats * present


def make_episode_fields(args, E, seed=0, terminated_p=0.9):
    """All transition fields of one rollout of E episodes ([E, T+1, nA, ...]), fp32 on CPU."""
    gen = torch.Generator().manual_seed(seed)
    T1 = args.episode_limit + 1
    nA, N, d = args.n_agents, args.max_vehicle_num, args.obs_shape_single
    Z, A, M = args.latent_dim, args.attention_dim, args.rnn_hidden_dim
    f = {}
    f["history"] = make_history(gen, (E, T1, nA), N, d)
    lat = -torch.log(torch.rand(E, T1, nA, N, Z, generator=gen).clamp_min(1e-12))
    f["behavior_latent"] = lat / lat.sum(-1, keepdim=True)          # Dirichlet(1) rows
    f["attention_latent"] = torch.randn(E, T1, nA, N, A, generator=gen) * 0.1
    f["rnn_states_actors"] = torch.randn(E, T1, nA, M, generator=gen) * 0.1
    f["rnn_states_critics"] = torch.randn(E, T1, nA, M, generator=gen) * 0.1
    f["reward"] = torch.randn(E, T1, nA, 1, generator=gen)
    f["terminated"] = (torch.rand(E, T1, nA, 1, generator=gen) < terminated_p).to(torch.uint8)
    f["actions"] = torch.randint(0, args.n_actions, (E, T1, nA, 1), generator=gen)
    f["actions_onehot"] = torch.nn.functional.one_hot(f["actions"][..., 0], args.n_actions).float()
    avail = (torch.rand(E, T1, nA, args.n_actions, generator=gen) < 0.8).int()
    avail[..., 0] = 1                                               # never all-zero per row
    f["avail_actions"] = avail
    f["obs"] = torch.zeros(E, T1, nA, args.obs_shape)
    f["state"] = torch.zeros(E, T1, args.state_shape)
    f["speed"] = torch.zeros(E, T1, nA, 1)
    f["filled"] = torch.ones(E, T1, 1, dtype=torch.long)
    return f


def make_batch(args, E, seed=0, terminated_p=0.9, device="cpu"):
    f = make_episode_fields(args, E, seed, terminated_p)
    if device != "cpu":
        f = {k: v.to(device) for k, v in f.items()}
    return DictBatch(f, E, args.episode_limit + 1, device)


def make_scheme(args):
    """The scheme dict run_ippo.py:160-180 builds (after ReplayBuffer preprocessing adds
    ``actions_onehot`` and ``filled``)."""
    return {
        "state": {"vshape": args.state_shape},
        "obs": {"vshape": args.obs_shape, "group": "agents"},
        "actions": {"vshape": (1,), "group": "agents", "dtype": torch.long},
        "rnn_states_actors": {"vshape": (args.rnn_hidden_dim,), "group": "agents"},
        "rnn_states_critics": {"vshape": (args.rnn_hidden_dim,), "group": "agents"},
        "history": {"vshape": (args.max_vehicle_num, args.obs_shape_single), "group": "agents"},
        "behavior_latent": {"vshape": (args.max_vehicle_num, args.latent_dim), "group": "agents"},
        "attention_latent": {"vshape": (args.max_vehicle_num, args.attention_dim), "group": "agents"},
        "avail_actions": {"vshape": (args.n_actions,), "group": "agents", "dtype": torch.int},
        "reward": {"vshape": (1,), "group": "agents"},
        "speed": {"vshape": (1,), "group": "agents"},
        "terminated": {"vshape": (1,), "group": "agents", "dtype": torch.uint8},
        "actions_onehot": {"vshape": (args.n_actions,), "group": "agents", "dtype": torch.float32},
        "filled": {"vshape": (1,), "dtype": torch.long},
    }


def rollout_step_inputs(args, E, seed=0):
    """Numpy inputs of one rollout vector step as ParallelRunner hands them over
    (runners/ippo_parallel_runner.py:218-231): history_single float64 [E,nA,N,d],
    history window float64 [E,nA,N,L,d]."""
    gen = torch.Generator().manual_seed(seed)
    nA, N, d, L = args.n_agents, args.max_vehicle_num, args.obs_shape_single, args.max_history_len
    hist_single = make_history(gen, (E, nA), N, d).double().numpy()
    window = make_history(gen, (E, nA, N), L, d)           # [E,nA,N,L,d]
    return hist_single, window.double().numpy()


def obs_stream(K, nA, obs_num, d, T, seed, n_ids=14, p_seen=0.6):
    """Synthetic Highway-style observation stream [T][K, nA, obs_num, 1 + d]: row 0 = the ego (id = agent index + 1),
    other rows = vehicles drawn from a pool of ids that come and go; unobserved rows are all-zero."""
    rng = np.random.default_rng(seed)
    steps = []
    for _ in range(T):
        obs = np.zeros((K, nA, obs_num, d + 1))
        for k in range(K):
            for i in range(nA):
                obs[k, i, 0, 0] = i + 1
                obs[k, i, 0, 1:] = rng.uniform(-1, 1, d)
                ids = rng.choice(np.arange(10, 10 + n_ids), size=obs_num - 1, replace=False)
                for j in range(1, obs_num):
                    if rng.random() < p_seen:
                        obs[k, i, j, 0] = ids[j - 1]
                        obs[k, i, j, 1:] = rng.uniform(-1, 1, d)
        steps.append(obs)
    return steps


class StubHighwayVecEnv:
    """Stand-in for envs/env_wrappers.SubprocVecEnv around Heterogeneous_Highway_Env (not installable offline) with the
    call surface ParallelRunner touches (runners/ippo_parallel_runner.py:90-102, 184): ``reset() -> (state, obs)``,
    ``step(action_env) -> (state, obs, reward, win_tags, terminated_agent, env_info)``, ``close()``.
    obs [K, nA, obs_num, 1 + d]: row 0 = the ego (id = agent index + 1), other rows = vehicles of an id pool that come and
    go; state [K, 1, n_state * (1 + d)].  The stream is pre-generated (it does not react to the actions -- the hot path never
    looks inside the simulator); ``end_steps[k]``: the step after which env k reports every agent terminated."""

    def __init__(self, K, nA, obs_num, d, n_state, T, seed=0, end_steps=None, n_ids=14, p_seen=0.6):
        rng = np.random.default_rng(seed)
        self.K, self.nA, self.T = K, nA, T
        obs = np.zeros((T + 2, K, nA, obs_num, d + 1))
        obs[:, :, :, 0, 0] = np.arange(1, nA + 1)
        obs[:, :, :, 0, 1:] = rng.uniform(-1, 1, (T + 2, K, nA, d))
        ids = 10 + np.argsort(rng.random((T + 2, K, nA, n_ids)), axis=-1)[..., :obs_num - 1]
        seen = rng.random((T + 2, K, nA, obs_num - 1)) < p_seen
        obs[..., 1:, 0] = np.where(seen, ids, 0)
        obs[..., 1:, 1:] = rng.uniform(-1, 1, (T + 2, K, nA, obs_num - 1, d)) * seen[..., None]
        self.obs = obs
        self.state = rng.uniform(-1, 1, (T + 2, K, 1, n_state * (d + 1)))
        self.reward = rng.normal(size=(T + 2, K, nA))
        self.end_steps = np.full(K, T + 5) if end_steps is None else np.asarray(end_steps)
        self.t = 0

    def shard(self, lo, hi):
        """envs [lo, hi) of this vector env as a vector env of their own (a data-parallel rank's share: the same streams the
        union's env would deliver for those rows)"""
        import copy
        out = copy.copy(self)
        out.K = hi - lo
        out.obs, out.state, out.reward = self.obs[:, lo:hi], self.state[:, lo:hi], self.reward[:, lo:hi]
        out.end_steps = self.end_steps[lo:hi]
        out.t = 0
        out._k0 = getattr(self, "_k0", 0) + lo
        return out

    def reset(self):
        self.t = 0
        return self.state[0], self.obs[0]

    def step(self, action_env):
        assert len(action_env) == self.K
        self.t += 1
        t = min(self.t, self.T + 1)
        term = np.repeat((self.t >= self.end_steps)[:, None], self.nA, axis=1)
        info = [{"speed": np.full(self.nA, 20.0 + k + getattr(self, "_k0", 0))} for k in range(self.K)]
        return self.state[t], self.obs[t], self.reward[t], np.zeros((self.K, self.nA)), term, info

    def close(self):
        pass
