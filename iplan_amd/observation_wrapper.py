"""id -> slot history wrapper of the Highway observations (reference: observation_wrapper.py:6-173, class
``observersation_state_history_wrapper``): the source of the ``[E, nA, N, L, d]`` history windows and
``[E, nA, N, d]`` single-step tensors the GAT / behaviour kernels consume.

The reference keeps, per (thread, agent), a Python list of the vehicle ids seen so far (slot = order of first
appearance) and one ``deque`` per slot, and walks threads x agents x observed rows in Python every step -- the top host
cost of a rollout at E >= 32.  Here the same state is three dense arrays

    slot_id [K, nA, N] int64 (-1 = free)     count [K, nA, N] int32     hist [K, nA, N, Tmax, d] float64

and a step is vectorised over (thread, agent); only the loop over the observed rows of one agent (obs_num, ~15) stays
sequential, because slot assignment depends on the order of appearance within a step.  Outputs are gathers.  Same
public surface and return values as the reference class (the dict/deque views are rebuilt on demand only)."""
from collections import deque

import numpy as np


class observersation_state_history_wrapper:                        # (sic) the reference's spelling
    def __init__(self, args, n_agents, max_vehicle_num, max_episode_len, max_history_len):
        self.args = args
        self.max_vehicle_num = max_vehicle_num
        self.max_episode_len = max_episode_len
        self.max_history_len = max_history_len
        self.obs_shape = args.obs_shape_single
        self.n_agents = n_agents
        self.n_threads = args.batch_size_run
        self.curr_t = 0
        self.history_out = None
        self.history_episode_out = None
        self._agent_ids = None

    # ------------------------------------------------------------------------------------------ state
    def agent_obs_profile_init(self, obs):
        """observation_wrapper.py:26-46: agent ids per thread in order of first appearance; empty histories."""
        obs = np.asarray(obs)
        K, nA = obs.shape[:2]
        N, Tm, d = self.max_vehicle_num, self.max_episode_len, self.obs_shape
        ids = obs[:, :, 0, 0].astype(np.int64)                      # [K, nA]
        # order of first appearance with duplicates dropped (a duplicated id keeps its first position)
        self._agent_ids = np.full((K, nA), np.iinfo(np.int64).min, dtype=np.int64)
        self._n_agent_ids = np.zeros(K, dtype=np.int64)
        for i in range(nA):                                         # nA is tiny; vectorised over threads
            seen = (self._agent_ids == ids[:, i:i + 1]).any(axis=1)
            rows = np.nonzero(~seen)[0]
            self._agent_ids[rows, self._n_agent_ids[rows]] = ids[rows, i]
            self._n_agent_ids[rows] += 1
        self._slot_id = np.full((K, nA, N), -1, dtype=np.int64)
        self._n_slots = np.zeros((K, nA), dtype=np.int64)
        self._count = np.zeros((K, nA, N), dtype=np.int64)
        self._hist = np.zeros((K, nA, N, Tm, d), dtype=np.float64)
        self._win = np.zeros((K, nA, N, self.max_history_len, d), dtype=np.float64)     # right-aligned last L entries, kept current
        return self.history

    def pure_obs_state_wrapper(self, state, obs):
        """observation_wrapper.py:52-60: drop the id column."""
        obs = np.asarray(obs)
        n_threads, n_agents, obs_num, obs_dim = obs.shape
        state_dim = state.shape[2]
        n_vehicles = int(state_dim // obs_dim)
        new_state = state.reshape(n_threads, 1, n_vehicles, obs_dim)[:, :, :, 1:].reshape((n_threads, -1))
        new_obs = obs[:, :, :, 1:].reshape((n_threads, n_agents, -1))
        return new_state, new_obs

    def _append(self, rows_k, rows_a, slots, values):
        """Append one entry to the deques (thread, agent, slot); a full deque drops its oldest entry (maxlen)."""
        Tm = self.max_episode_len
        cnt = self._count[rows_k, rows_a, slots]
        full = cnt >= Tm
        if full.any():
            fk, fa, fs = rows_k[full], rows_a[full], slots[full]
            self._hist[fk, fa, fs, :-1] = self._hist[fk, fa, fs, 1:]
            cnt = np.where(full, Tm - 1, cnt)
        self._hist[rows_k, rows_a, slots, cnt] = values
        self._count[rows_k, rows_a, slots] = cnt + 1
        w = self._win[rows_k, rows_a, slots]
        w[:, :-1] = w[:, 1:]
        w[:, -1] = values
        self._win[rows_k, rows_a, slots] = w

    def obs_history_create(self, obs):
        """observation_wrapper.py:68-97.  Observed rows (any non-zero entry) go to the slot of their id (new ids take
        the next free slot, in row order); every known id that was not observed this step gets a zero entry."""
        obs = np.asarray(obs, dtype=np.float64)
        K, nA, obs_num, obs_dim = obs.shape
        N = self.max_vehicle_num
        ego = obs[:, :, 0, 0].astype(np.int64)
        hit = self._agent_ids[:, None, :] == ego[:, :, None]
        if not hit.any(axis=2).all():                                                      # list.index raises in the reference
            k, a = np.argwhere(~hit.any(axis=2))[0]
            raise ValueError(f"{int(ego[k, a])} is not in list (ego id of thread {k}, agent {a} was never registered by "
                             "agent_obs_profile_init)")
        agent_idx = hit.argmax(axis=2)                                                    # self.agent_id[k].index(agent_id)
        kk = np.repeat(np.arange(K), nA)
        aa = agent_idx.reshape(-1)
        seen = np.zeros((K * nA, N), dtype=bool)                     # slots this (thread, agent) pair observed in this step
        order = np.arange(K * nA)
        for j in range(obs_num):
            row = obs[:, :, j, :].reshape(K * nA, obs_dim)
            present = row.any(axis=1)
            vid = row[:, 0].astype(np.int64)
            # two agents of a thread resolving to the same agent_idx (duplicate ego ids; not produced by the simulator)
            # are processed one after the other, like the reference: conflict-free groups, almost always a single one
            for grp in _conflict_free_groups(kk, aa, present, order):
                gk, ga = kk[grp], aa[grp]
                match = self._slot_id[gk, ga] == vid[grp, None]                           # [g, N]
                has = match.any(axis=1)
                slot = np.where(has, match.argmax(axis=1), self._n_slots[gk, ga])
                if (slot >= N).any():
                    raise IndexError("more than max_vehicle_num vehicles observed by one agent")
                new = ~has
                self._slot_id[gk[new], ga[new], slot[new]] = vid[grp][new]
                self._n_slots[gk[new], ga[new]] += 1
                self._append(gk, ga, slot, row[grp, 1:])
                seen[grp, slot] = True
        # known ids the pair did not observe in this step: a zero entry each (:92-96)
        for grp in _conflict_free_groups(kk, aa, np.ones(K * nA, dtype=bool), order):
            gk, ga = kk[grp], aa[grp]
            exist = np.arange(N)[None, :] < self._n_slots[gk, ga][:, None]
            r, s = np.nonzero(exist & ~seen[grp])
            if len(r):
                self._append(gk[r], ga[r], s, np.zeros((len(r), obs_dim - 1)))
        # the reference returns its bookkeeping objects here; callers rarely look at them, so they are built on first use
        return _Lazy(lambda: self.agent_id), _Lazy(lambda: self.obs_vehicle_id), _Lazy(lambda: self.history)

    # ------------------------------------------------------------------------------------------ outputs
    def _tail(self, length):
        """Right-aligned last `length` entries of every deque: [K, nA, N, length, d]."""
        cnt = self._count[..., None]                                                       # [K, nA, N, 1]
        pos = cnt - length + np.arange(length)                                             # entry index per output column
        ok = pos >= 0
        g = np.take_along_axis(self._hist, np.clip(pos, 0, None)[..., None], axis=3)
        return np.where(ok[..., None], g, 0.0)

    def obs_history_output(self):
        """observation_wrapper.py:101-120 -> [K, nA, N, L, d]."""
        self.history_out = self._win.copy()
        return self.history_out

    def obs_single_history_output(self):
        """observation_wrapper.py:125-141 -> [K, nA, N, d]: the latest entry of every slot."""
        self.single_history_out = self._win[:, :, :, -1, :].copy()
        return self.single_history_out

    def obs_history_episode_output(self, mask):
        """observation_wrapper.py:145-173: the whole (masked) episode history and its [.., T // L, L, d] view."""
        mask = np.asarray(mask)                                                            # [K, T, nA]
        Tm, L = self.max_episode_len, self.max_history_len
        raw = self._tail(Tm)
        cnt = self._count[..., None]
        col = np.arange(Tm)
        filled = col >= Tm - cnt                                                           # columns the reference writes
        m = mask[:, :Tm].transpose(0, 2, 1)[:, :, None, :]                                 # [K, nA, 1, T]
        raw = np.where(filled[..., None], raw * m[..., None], 0.0)
        self.raw_history_episode_out = raw
        self.history_episode_out = raw.reshape((self.n_threads, self.n_agents, self.max_vehicle_num, int(Tm // L), L, self.obs_shape))
        return self.raw_history_episode_out, self.history_episode_out

    # ------------------------------------------------------------------------------------------ reference-shaped views
    @property
    def agent_id(self):
        return None if self._agent_ids is None else [list(self._agent_ids[k, :self._n_agent_ids[k]]) for k in range(len(self._agent_ids))]

    @property
    def obs_vehicle_id(self):
        K, nA = self._n_slots.shape
        return [[list(self._slot_id[k, i, :self._n_slots[k, i]]) for i in range(nA)] for k in range(K)]

    @property
    def history(self):
        if self._agent_ids is None:
            return None
        K, nA = self._n_slots.shape
        out = []
        for k in range(K):
            per = {}
            for i in range(nA):
                per[i] = {s: deque((self._hist[k, i, s, t].copy() for t in range(self._count[k, i, s])), maxlen=self.max_episode_len)
                          for s in range(self._n_slots[k, i])}
            out.append(per)
        return out


class _Lazy:
    """List / dict stand-in that materialises the reference-shaped object (lists of ids, dict of deques) on first use."""

    def __init__(self, build):
        self._build, self._obj = build, None

    def _get(self):
        if self._obj is None:
            self._obj = self._build()
        return self._obj

    def __getitem__(self, k):
        return self._get()[k]

    def __len__(self):
        return len(self._get())

    def __iter__(self):
        return iter(self._get())

    def __eq__(self, other):
        return self._get() == (other._get() if isinstance(other, _Lazy) else other)

    def __repr__(self):
        return repr(self._get())


def _conflict_free_groups(kk, aa, active, order):
    """Split the active (thread, agent) pairs into groups in which no two pairs address the same (thread, agent_idx)
    state, preserving agent order between groups (almost always a single group)."""
    idx = order[active]
    if len(idx) == 0:
        return []
    key = kk[idx] * (aa.max() + 1) + aa[idx]
    groups = []
    remaining = idx
    rem_key = key
    while len(remaining):
        _, first = np.unique(rem_key, return_index=True)
        first.sort()
        groups.append(remaining[first])
        keep = np.ones(len(remaining), dtype=bool)
        keep[first] = False
        remaining, rem_key = remaining[keep], rem_key[keep]
    return groups


class DeviceObsHistory:
    """The same history, device-resident (csrc/obs_history.hip: iplan_obs_history_step): what ParallelRunner
    uses on a GPU instead of the numpy class above.  Per environment step ONE host -> device copy of the raw observations
    ([K, nA, obs_num, 1 + d] as float32 through a pinned buffer) and ONE launch; ``win`` ([K, nA, N, L, d] float32) IS
    obs_history_output() and the single-step view is written wherever the caller points (the episode container's history field).
    Values are copies: bit-identical to the numpy class after its float32 cast (tests/test_obs_wrapper.py).  Errors the numpy class
    raises where they occur (unregistered ego id: ValueError; more than max_vehicle_num vehicles: IndexError) are flagged on the
    device and raised by ``check()``, which the runner calls at its per-step host synchronisation."""

    def __init__(self, n_threads, n_agents, max_vehicle_num, max_history_len, obs_dim, device):
        import torch
        self.K, self.nA, self.N, self.L, self.d = n_threads, n_agents, max_vehicle_num, max_history_len, obs_dim
        self.device = torch.device(device)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.slot_id = torch.full((self.K, self.nA, self.N), -1, **i32)
        self.n_slots = torch.zeros(self.K, self.nA, **i32)
        self.agent_ids = torch.zeros(self.K, self.nA, **i32)
        self.win = torch.zeros(self.K, self.nA, self.N, self.L, self.d, dtype=torch.float32, device=self.device)
        self.err = torch.zeros(1, **i32)
        self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory() if self.device.type == "cuda" else None
        self._obs_pin = None
        self._obs_dev = None

    def init(self, obs):
        """agent_obs_profile_init (observation_wrapper.py:26-46): registered ego ids per thread in order of first appearance,
        empty histories.  Once per episode: on the host."""
        import torch
        obs = np.asarray(obs)
        K, nA = obs.shape[:2]
        assert (K, nA) == (self.K, self.nA), (obs.shape, self.K, self.nA)
        ids = obs[:, :, 0, 0].astype(np.int64)
        reg = np.full((K, nA), np.iinfo(np.int32).min, dtype=np.int64)
        n_reg = np.zeros(K, dtype=np.int64)
        for i in range(nA):
            seen = (reg == ids[:, i:i + 1]).any(axis=1)
            rows = np.nonzero(~seen)[0]
            reg[rows, n_reg[rows]] = ids[rows, i]
            n_reg[rows] += 1
        self.agent_ids.copy_(torch.as_tensor(reg.astype(np.int32)))
        self.slot_id.fill_(-1)
        self.n_slots.zero_()
        self.win.zero_()
        self.err.zero_()

    def step(self, obs, single_out=None):
        """obs_history_create + the two outputs.  ``single_out``: a [K, nA, N, d] float32 view (last two dims contiguous) that
        receives obs_single_history_output(); returns ``win`` (= obs_history_output(), valid until the next step())."""
        import torch
        from . import _lib as L
        obs = np.asarray(obs)
        K, nA, obs_num, W = obs.shape
        assert (K, nA, W) == (self.K, self.nA, self.d + 1), obs.shape
        if self.device.type == "cuda":
            if self._obs_pin is None or self._obs_pin[0].shape != obs.shape:
                t = torch.empty(obs.shape, dtype=torch.float32, pin_memory=True)
                self._obs_pin = (t, t.numpy())
                self._obs_dev = torch.empty(obs.shape, dtype=torch.float32, device=self.device)
            np.copyto(self._obs_pin[1], obs, casting="unsafe")
            self._obs_dev.copy_(self._obs_pin[0], non_blocking=True)
            od = self._obs_dev
        else:
            od = torch.as_tensor(obs, dtype=torch.float32).contiguous()
            self._keep = od
        a = L.ObsHistArgs()
        a.K, a.nA, a.N, a.L, a.d, a.obs_num = K, nA, self.N, self.L, self.d, obs_num
        a.obs, a.agent_ids, a.slot_id, a.n_slots = od.data_ptr(), self.agent_ids.data_ptr(), self.slot_id.data_ptr(), self.n_slots.data_ptr()
        a.win, a.err = self.win.data_ptr(), self.err.data_ptr()
        if single_out is not None:
            assert single_out.shape == (K, nA, self.N, self.d) and single_out.dtype == torch.float32
            assert single_out.stride(3) == 1 and single_out.stride(2) == self.d, single_out.stride()
            a.single, a.single_s_k, a.single_s_a = single_out.data_ptr(), single_out.stride(0), single_out.stride(1)
        L.get_lib().call("iplan_obs_history_step", a, L.current_stream(self.device))
        return self.win

    def stage_error_flag(self):
        """enqueue the copy of the error word to the host (read it with check() after the next stream synchronise)"""
        if self._err_host is not None:
            self._err_host.copy_(self.err, non_blocking=True)

    def check(self):
        flag = int(self._err_host[0]) if self._err_host is not None else int(self.err[0])
        if flag & 1:
            raise ValueError("an ego id is not in list (it was never registered by agent_obs_profile_init)")
        if flag & 2:
            raise IndexError("more than max_vehicle_num vehicles observed by one agent")
