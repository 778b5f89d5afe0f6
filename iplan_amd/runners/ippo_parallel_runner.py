"""ParallelRunner -- device-resident drop-in for runners/ippo_parallel_runner.py:7-301.

Same constructor / ``setup`` / ``run`` / ``reset`` / ``get_env_info`` / ``close_env`` surface, same order of calls inside
``run`` (select_actions_ippo -> env.step -> obs history -> GAT_latent_update -> latent_update -> EpisodeBatch.update),
same return value ``(batch, avg_win_rates, avg_rwd, avg_len)``.  What differs is where the data lives:

  * the episode batch is created on ``args.device`` and the three fused launches of a vector step read their inputs from it
    and write their outputs (actions + one-hot, GRU states, attention / behaviour latents) into it IN PLACE -- the
    reference round-trips all of them through numpy and ``EpisodeBatch.update`` every step (~12 H<->D copies);
  * per step the host sees exactly one device->host copy (the ``[E, nA]`` action indices the simulator needs, pinned) and
    one batch of host->device copies of what the simulator produced (entity observations / history window, reward,
    terminated, state, obs; pinned, asynchronous);
  * the id -> slot history wrapper is iplan_amd.observation_wrapper (vectorised; identical outputs).

The episode container is the caller's own ``components.episode_buffer.EpisodeBatch`` when that module is importable (the
reference tree is on sys.path: run_ippo.py drops in unchanged), otherwise iplan_amd.synth.DictBatch with the same fields.
"""
from functools import partial

import numpy as np
import torch

from .. import ops, synth
from ..observation_wrapper import DeviceObsHistory, observersation_state_history_wrapper


def _dict_batch(scheme, groups, batch_size, max_seq_length, preprocess=None, device="cpu"):
    """EpisodeBatch-shaped fallback: zero-initialised fields of ``scheme`` (+ the preprocess outputs and ``filled``)."""
    data = {}
    full = dict(scheme)
    full.setdefault("filled", {"vshape": (1,), "dtype": torch.long})
    for k, (new_k, transforms) in (preprocess or {}).items():
        vshape, dtype = full[k]["vshape"], full[k].get("dtype", torch.float32)
        for t in transforms:
            vshape, dtype = t.infer_output_info(vshape, dtype)
        full[new_k] = {"vshape": vshape, "dtype": dtype, **({"group": full[k]["group"]} if "group" in full[k] else {})}
    for k, info in full.items():
        vshape = info["vshape"]
        vshape = (vshape,) if isinstance(vshape, int) else tuple(vshape)
        grp = (groups[info["group"]],) if "group" in info else ()
        data[k] = torch.zeros((batch_size, max_seq_length) + grp + vshape, dtype=info.get("dtype", torch.float32), device=device)
    return synth.DictBatch(data, batch_size, max_seq_length, device)


class ParallelRunner:
    def __init__(self, args, env, logger):
        self.args = args
        self.logger = logger
        self.batch_size = self.args.batch_size_run
        self.env = env
        self.episode_limit = self.args.episode_length if args.env == "MPE" else self.args.episode_limit
        self.t = 0
        self.t_env = 0
        self.train_returns, self.test_returns = [], []
        self.train_stats, self.test_stats = {}, {}
        self.log_train_stats_t = -1000000
        self.max_vehicle_num = args.n_other_vehicles + args.n_agents
        self.n_agents = args.n_agents
        self.episode_length = args.episode_limit
        self.history_wrapper = observersation_state_history_wrapper(args, self.n_agents, self.max_vehicle_num,
                                                                    self.episode_length, args.max_history_len)
        self.device = torch.device(getattr(args, "device", "cuda" if args.use_cuda else "cpu"))
        self._pin = {}
        self._dev_history = None
        self.host_seconds = 0.0        # time spent outside the device work (env.step + history wrapper), for reporting
        # Data-parallel runs (iplan_amd.parallel.DataParallel.attach(..., runner=self)): this runner steps ITS shard of the parallel
        # environments (args.batch_size_run = the rank's share); the rollout itself needs no communication -- no env reads another's
        # data anywhere on the path -- but ``t_env`` drives the schedules every rank must take identically (Behavior_warmup /
        # GAT_warmup gates and the save / log intervals of run_ippo.py:268-311, the linear lr decay of learners/ippo_learner.py:86-91):
        # it counts the environment steps of ALL ranks, and the logged episode averages are those of the union.
        self.dp = None

    def setup(self, scheme, groups, preprocess, mac, behavior_learner, prediction_learner):
        try:
            from components.episode_buffer import EpisodeBatch          # the caller's pymarl container, if present
            factory = EpisodeBatch
        except ImportError:
            factory = _dict_batch
        self.new_batch = partial(factory, scheme, groups, self.batch_size, self.episode_limit + 1, preprocess=preprocess,
                                 device=self.device)
        self.mac = mac
        self.scheme, self.groups, self.preprocess = scheme, groups, preprocess
        self.behavior_learner = behavior_learner
        self.prediction_learner = prediction_learner

    def get_env_info(self, args):
        return {"n_agents": self.n_agents, "n_actions": args.n_actions, "state_shape": args.obs_shape_single * self.max_vehicle_num,
                "episode_limit": self.episode_length, "obs_shape": args.obs_shape_single * args.n_obs_vehicles}

    def close_env(self):
        self.env.close()

    def reset(self):
        self.batch = self.new_batch()
        state, obs = self.env.reset()
        self.t = 0
        self.env_steps_this_run = 0
        return state, obs

    # ------------------------------------------------------------------------------------------ host <-> device plumbing
    def _to_dev(self, name, array, dtype):
        """numpy -> device through a pinned staging buffer (asynchronous on the current stream).  The dtype conversion is a
        single-threaded numpy copy straight into the pinned pages (a torch copy_ of a few hundred KB fans out over every host
        core: 0.9 ms per call on a 256-core box, profiles/history/r02b_runner_host_in_loop.txt)."""
        a = np.asarray(array)
        if self.device.type != "cuda":
            return torch.as_tensor(a).to(dtype)
        buf = self._pin.get(name)
        if buf is None or buf[0].shape != a.shape or buf[0].dtype != dtype:
            t = torch.empty(a.shape, dtype=dtype, pin_memory=True)
            buf = self._pin[name] = (t, t.numpy())
        np.copyto(buf[1], a, casting="unsafe")
        return buf[0].to(self.device, non_blocking=True)

    def _field(self, key):
        return self.batch[key]                                          # [E, T1, ...] tensor of the episode container

    def _masked(self, array, shape, alive):
        out = np.zeros(shape)
        out[alive] = np.asarray(array).reshape(shape)[alive]
        return out

    # ------------------------------------------------------------------------------------------ one vectorised episode
    @torch.no_grad()
    def run(self, test_mode=False, noise=None, q_all=None):
        """``noise`` [T+1, nA, E, N, N-1, 2] / ``q_all`` [T, nA, E, n_actions]: pre-drawn gumbel samples (step t's update uses
        ``noise[t]``, the episode-initial one ``noise[T]``) and Exp(1) samples of the action race -- the parity tests inject
        them into this loop and into the oracle (same convention as harness.SyntheticLoop._rollout_body); default: drawn on
        the device per step, like the reference draws them inside F.gumbel_softmax / Categorical.sample."""
        import os
        import time
        a, E, nA = self.args, self.args.batch_size_run, self.n_agents
        state, obs = self.reset()
        D = self._field
        dev = self.device
        episode_returns, episode_lengths, episode_wins = np.zeros(E), np.zeros(E), np.zeros(E)
        terminated = np.zeros(E, dtype=bool)
        alive = np.arange(E)
        hw = self.history_wrapper
        # The id -> slot history: on a GPU it lives in HBM (observation_wrapper.DeviceObsHistory, csrc/obs_history.hip: one 58 KB
        # copy of the raw observations + one launch per step, the single-step view written straight into the episode container, the
        # L-step windows read in place by the encoder); the numpy class does the same on the host (CPU runs, IPLAN_HOST_HISTORY=1).
        dh = None
        if dev.type == "cuda" and not os.environ.get("IPLAN_HOST_HISTORY"):
            dh = self._dev_history
            if dh is None or dh.K != E:
                dh = self._dev_history = DeviceObsHistory(E, nA, self.max_vehicle_num, a.max_history_len, a.obs_shape_single, dev)
            dh.init(obs)
            dh.step(obs, single_out=D("history")[:, 0])
        else:
            hw.agent_obs_profile_init(obs)
            hw.obs_history_create(obs)
            single = hw.obs_single_history_output()
        state, obs = hw.pure_obs_state_wrapper(state, obs)
        D("avail_actions").fill_(1)                                     # the reference stores all-ones every step (:108, 134)
        if dh is None:
            D("history")[:, 0] = self._to_dev("single", single, torch.float32)
        D("state")[:, 0] = self._to_dev("state", self._masked(state, (E, a.state_shape), alive), torch.float32)
        D("obs")[:, 0] = self._to_dev("obs", self._masked(obs, (E, nA, a.obs_shape), alive), torch.float32)
        for k in ("rnn_states_actors", "rnn_states_critics", "behavior_latent", "attention_latent"):
            D(k)[:, 0].zero_()
        eh = torch.zeros(2, E, a.num_encoder_layer, nA, self.max_vehicle_num, a.encoder_rnn_dim, device=dev)    # ping-pong
        if a.GAT_enable:
            self.prediction_learner.GAT_latent_update(D("history")[:, 0], D("attention_latent")[:, 0], D("behavior_latent")[:, 0],
                                                      out=D("attention_latent")[:, 0], noise=None if noise is None else noise[a.episode_limit])
        D("filled")[:, 0] = 1
        act_host = torch.empty(E, nA, dtype=torch.long, pin_memory=dev.type == "cuda")
        import os
        # the next step's action selection reads exactly what this step's latent updates write and the simulator only steps after
        # it: with both incentives on it rides in their launch (iplan_gat_enc_ac_fwd).  IPLAN_NO_FUSE_AC=1: its own launch
        fuse_ac = a.GAT_enable and a.Behavior_enable and not os.environ.get("IPLAN_NO_FUSE_AC") and E <= 512
        ac_in_flight = False
        sync_host = None
        for _ in range(a.episode_limit):
            t = self.t
            # actions, their one-hot and the new GRU states go straight into the episode container
            # (the reference never forwards test_mode to the controller -- it samples in test runs too, :172-173)
            if not ac_in_flight:           # (otherwise it rode in the previous step's latent-update launch, see below)
                self.mac.select_actions_ippo(self.batch, t_ep=t, as_numpy=False, write_back=True, q_noise=None if q_all is None else q_all[t])
            ac_in_flight = False
            if terminated.any():                 # envs that terminated earlier store action 0 (action2env_tuple, :81-83, 177-180)
                dead = torch.as_tensor(np.flatnonzero(terminated), device=dev)
                D("actions")[dead, t] = 0
                D("actions_onehot")[dead, t] = 0
                D("actions_onehot")[dead, t, :, 0] = 1
            act_host.copy_(D("actions")[:, t, :, 0], non_blocking=True)    # the ONE device -> host copy of the step
            if dh is not None:
                dh.stage_error_flag()
            if fuse_ac and dev.type == "cuda":
                # ... plus the fused launch's give-up flag (4 bytes): an action selection that stopped waiting for its launch's latent
                # updates must abort the episode BEFORE env.step sees its actions, not at the episode's end
                if sync_host is None:
                    sync_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                sync_host.copy_(ops._fused_sync(dev)[2:3], non_blocking=True)
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()
            if sync_host is not None and int(sync_host[0]) != 0:
                ops.check_fused_sync()                       # raises
            if dh is not None:
                dh.check()                                   # raises what the reference's wrapper raises (unknown ego id / too many vehicles)
            t0 = time.perf_counter()
            actions = act_host.numpy()
            action_env = [tuple(row) for row in actions.astype(np.float64)]
            state, obs, reward, win_tags, terminated_agent, env_info = self.env.step(action_env)
            reward, terminated_agent = np.asarray(reward), np.asarray(terminated_agent)
            speed = np.zeros((E, nA))
            if a.env == "highway":
                for i in range(E):
                    speed[i, :] = env_info[i]["speed"]
            terminated = np.logical_or(terminated_agent.all(axis=1), terminated)
            episode_returns += reward.sum(axis=1)
            episode_lengths += 1 - terminated
            episode_wins = np.asarray(win_tags).sum(axis=1)
            if not test_mode:
                self.env_steps_this_run += int((1 - terminated).sum())
            if terminated.all():
                self.host_seconds += time.perf_counter() - t0
                # the reference breaks before it stores the step's new GRU states (:212-214 vs :253-266); the fused launch
                # above has already written them at t + 1
                D("rnn_states_actors")[:, t + 1].zero_()
                D("rnn_states_critics")[:, t + 1].zero_()
                break
            if dh is not None:
                window_dev = dh.step(obs, single_out=D("history")[:, t + 1])
            else:
                hw.obs_history_create(obs)
                single = hw.obs_single_history_output()
                window = hw.obs_history_output() if a.Behavior_enable else None
            state, obs = hw.pure_obs_state_wrapper(state, obs)
            self.host_seconds += time.perf_counter() - t0
            # what the simulator produced: one batch of pinned host -> device copies
            if dh is None:
                D("history")[:, t + 1] = self._to_dev("single", single, torch.float32)
                window_dev = self._to_dev("window", window, torch.float32) if a.Behavior_enable else None
            if a.GAT_enable and a.Behavior_enable:           # both latent updates of the step in one launch (iplan_gat_enc_fwd)
                enc = self.behavior_learner.latent_update(window_dev, eh[t & 1], D("behavior_latent")[:, t],
                                                          out_latent=D("behavior_latent")[:, t + 1], out_hidden=eh[(t + 1) & 1][:, 0], launch=False)
                nxt = None
                step_noise = None if noise is None else noise[t]
                if fuse_ac and t + 1 < a.episode_limit:
                    if step_noise is None:       # random draws in the two-launch order: this update's gumbel samples, then the action race's
                        from ..nova.GAT_Net import gumbel_noise
                        step_noise = gumbel_noise((nA, E, self.max_vehicle_num, self.max_vehicle_num - 1, 2), dev)
                    nxt = self.mac.select_actions_ippo(self.batch, t_ep=t + 1, as_numpy=False, write_back=True, launch=False,
                                                       q_noise=None if q_all is None else q_all[t + 1])
                    ac_in_flight = True
                self.prediction_learner.GAT_latent_update(D("history")[:, t + 1], D("attention_latent")[:, t], D("behavior_latent")[:, t],
                                                          out=D("attention_latent")[:, t + 1], fuse_enc=enc, fuse_ac=nxt, noise=step_noise)
            elif a.GAT_enable:
                self.prediction_learner.GAT_latent_update(D("history")[:, t + 1], D("attention_latent")[:, t], D("behavior_latent")[:, t],
                                                          out=D("attention_latent")[:, t + 1], noise=None if noise is None else noise[t])
            elif a.Behavior_enable:
                self.behavior_learner.latent_update(window_dev, eh[t & 1], D("behavior_latent")[:, t],
                                                    out_latent=D("behavior_latent")[:, t + 1], out_hidden=eh[(t + 1) & 1][:, 0])
            D("reward")[:, t] = self._to_dev("reward", self._masked(reward, (E, nA), alive), torch.float32).unsqueeze(-1)
            D("terminated")[:, t] = self._to_dev("terminated", terminated_agent, torch.uint8).unsqueeze(-1)
            if "speed" in self.scheme:
                D("speed")[:, t] = self._to_dev("speed", speed, torch.float32).unsqueeze(-1)
            D("state")[:, t + 1] = self._to_dev("state", self._masked(state, (E, a.state_shape), alive), torch.float32)
            D("obs")[:, t + 1] = self._to_dev("obs", self._masked(obs, (E, nA, a.obs_shape), alive), torch.float32)
            self.t += 1
            D("filled")[:, self.t] = 1
            alive = np.flatnonzero(~terminated)
        if dev.type == "cuda":
            # the last step's host -> device copies read the pinned staging buffers asynchronously and nothing after them
            # synchronises: the next run()'s reset() would overwrite 'single' / 'state' / 'obs' while they are in flight
            if dh is not None:
                dh.stage_error_flag()                    # (the episode's LAST history step: the next run()'s init() would clear its flag)
            torch.cuda.current_stream(dev).synchronize()
        if dh is not None:
            # Where the raise lands: the history kernel of step t runs behind env.step(t) on the device and its error word reaches
            # the host with the NEXT per-step synchronisation (or here, for the last step) -- one step after the reference's wrapper,
            # which raises inside obs_history_create of step t itself.  The episode is aborted either way before its batch is used.
            dh.check()
        if fuse_ac:
            ops.check_fused_sync()
        steps_this_run = self.env_steps_this_run
        if self.dp is not None and self.dp.world > 1:
            # ONE small all-reduce per episode, after the rollout: [env steps, sum of wins / returns / lengths, envs] over the ranks
            tot = torch.tensor([float(steps_this_run), float(np.sum(episode_wins)), float(np.sum(episode_returns)),
                                float(np.sum(episode_lengths)), float(E)], dtype=torch.float64, device=dev)
            tot = self.dp.all_reduce_sum(tot).cpu().numpy()
            steps_this_run = int(round(tot[0]))
            avg_win_rates, avg_rwd, avg_len = tot[1] / tot[4], tot[2] / tot[4], tot[3] / tot[4]
        else:
            avg_win_rates, avg_rwd, avg_len = np.mean(episode_wins, axis=0), np.mean(episode_returns, axis=0), np.mean(episode_lengths, axis=0)
        if not test_mode:
            self.t_env += steps_this_run
        self._log(avg_win_rates, avg_rwd, avg_len)
        self.log_train_stats_t = self.t_env
        return self.batch, avg_win_rates, avg_rwd, avg_len

    def _log(self, win_rates, episode_reward, episode_len):
        self.logger.log_stat(self.args.log_prefix + "Average episode_win_num", win_rates, self.t_env)
        self.logger.log_stat(self.args.log_prefix + "Average episode_reward", episode_reward, self.t_env)
        self.logger.log_stat(self.args.log_prefix + "Average episode_len", episode_len, self.t_env)
