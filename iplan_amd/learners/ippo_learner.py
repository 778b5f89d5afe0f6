"""IPPOLearner -- independent PPO over the decentralised controller (mirror of
learners/ippo_learner.py:17-424).

Same constructor, ``insert_episode_batch`` / ``train`` / ``save_models`` / ``load_models`` / ``lr_decay``
contracts, same optimiser checkpoint files.  What differs is the execution plan:

  * episodes are kept in ONE device-resident store ``[buffer_size, T+1, n_agents, ...]`` per field
    (the reference keeps per-agent deques of per-episode tensors and ``th.cat``s them at train time);
  * the reference trains agent after agent (15 epochs each); agents own private networks and
    private data, so the updates are independent and here every PPO epoch is ONE fused
    forward / loss / backward / clip+Adam launch sequence covering all agents, actors and critics;
  * the ``[256, 91, F]`` input tensor of ``_build_inputs_ippo`` is never materialised -- the kernels
    gather features from the stored fields; the ``randperm`` minibatch shuffle is dropped because
    ``num_mini_batch == 1`` makes every epoch a full-batch step (row order only permutes a sum).

Host code here is bookkeeping only: every number is produced by kernels in libiplan_hip.so.
"""
import copy

import os

import torch as th

from .. import _lib as L
from .. import ops
from ..optim import FusedAdam, step_all
from ..streams import AsyncHost
from ..utils.mappo_utils.util import update_linear_schedule

_STORE_KEYS = ("history", "attention_latent", "behavior_latent", "actions", "avail_actions", "reward",
               "terminated", "rnn_states_actors", "rnn_states_critics", "actions_onehot", "obs", "state")


class EpisodeStore:
    """Device-resident stand-in for the n_agents ``SeparatedReplayBuffer`` deques
    (utils/mappo_utils/separated_buffer.py:14-110): keeps the most recent ``buffer_size`` episodes
    in insertion order."""

    def __init__(self, args, device):
        self.size = args.buffer_size
        self.device = device
        self.count = 0
        self.data = {}
        self.consumed = set()                       # agents whose per-agent view was cleared since the last insert / clear

    def insert(self, ep_batch, n_eps):
        self.consumed.clear()
        for key in _STORE_KEYS:
            try:
                src = ep_batch[key]
            except (KeyError, ValueError):
                continue
            src = src[:n_eps]
            if key not in self.data:
                self.data[key] = th.zeros((self.size,) + tuple(src.shape[1:]), dtype=src.dtype, device=self.device)
        over = self.count + n_eps - self.size
        if over > 0:                                    # deque(maxlen) semantics: drop the oldest
            keep = self.count - over
            for v in self.data.values():
                if keep > 0:
                    v[:keep] = v[over:self.count].clone()
            self.count = max(keep, 0)
        take = min(n_eps, self.size)
        for key, v in self.data.items():
            v[self.count:self.count + take] = ep_batch[key][n_eps - take:n_eps].to(self.device)
        self.count += take

    def clear(self):
        self.count = 0
        self.consumed.clear()


class _AgentBuffer:
    """Per-agent view with the SeparatedReplayBuffer surface callers touch.  The n_agents views share one EpisodeStore;
    ``clear_buffer`` marks only THIS agent's view consumed (the reference clears one agent's deque,
    separated_buffer.py:44-50) and the store is reset once every agent has cleared -- so the reference-shaped per-agent
    loop (get_batch / ppo_update / clear_buffer, agent after agent) sees every agent's data."""

    def __init__(self, store, agent, n_agents):
        self.store, self.agent, self.n_agents = store, agent, n_agents

    def can_sample(self):
        return self.store.count == self.store.size and self.agent not in self.store.consumed

    def clear_buffer(self):
        self.store.consumed.add(self.agent)
        if len(self.store.consumed) >= self.n_agents:
            self.store.clear()

    def get_batch(self):
        if not self.can_sample():
            return None
        d, i = self.store.data, self.agent
        out = {"actions": d["actions"][:, :, i], "actions_onehot": d["actions_onehot"][:, :, i],
               "rnn_states_actor": d["rnn_states_actors"][:, :, i], "rnn_states_critic": d["rnn_states_critics"][:, :, i],
               "reward": d["reward"][:, :, i], "terminated_masks": 1 - d["terminated"][:, :, i],
               "history": d["history"][:, :, i], "available_actions": d["avail_actions"][:, :, i]}
        if "obs" in d:
            out["obs"] = d["obs"][:, :, i]
        if "state" in d:
            out["state"] = d["state"]
        for k in ("behavior_latent", "attention_latent"):
            if k in d:
                out[k] = d[k][:, :, i]
        return out


class IPPOLearner:
    def __init__(self, mac, scheme, logger, args):
        self.device = th.device("cuda" if args.use_cuda else "cpu")
        self.args = args
        self.lr = args.lr
        self.critic_lr = args.critic_lr
        self.use_linear_lr_decay = args.use_linear_lr_decay
        self.optim_eps = args.optim_eps
        self.weight_decay = args.weight_decay
        self.tpdv = dict(dtype=th.float32, device=self.device)
        self.n_agents = args.n_agents
        self.t_max = args.t_max
        self.episode_limit = args.episode_limit
        self.batch_size_run = args.batch_size_run
        self.batch_size = args.batch_size
        self.mac = mac
        self.state_shape = self.mac.input_scheme["state"]["vshape"]
        self.obs_shape = self.mac.input_scheme["obs"]["vshape"]
        self.n_actions = args.n_actions
        self.logger = logger
        self.log_prefix = args.log_prefix
        self.log_stats_t = -self.args.learner_log_interval - 1

        self.store = EpisodeStore(args, self.device)
        self.buffers = [_AgentBuffer(self.store, i, self.n_agents) for i in range(self.n_agents)]
        # diagnostics / parity tests: keep a copy of both parameter arenas as they were BEFORE train()'s last optimiser
        # step (``last_step_params``), so that the last step's gradients can be checked at that exact parameter point
        self.probe_last_step = False
        self.last_step_params = None
        self.last_step_moments = None                        # ((exp_avg, exp_avg_sq, steps taken) of the actor arena, same of the critic arena)
        self.last_step_relu = None                           # ([2, nA, rows, M] bool, same): which side of the fc1 / fc2 ReLU kink each unit took
        # ... and the same at EARLIER optimiser steps of one train(): ``probe_steps`` = step indices (0-based) whose complete
        # state -- parameters, moments and ReLU branches before the step, clipped gradients and parameters after it -- goes to
        # ``step_probes[k]``; ``probe_all_adam`` keeps parameters / moments / clipped gradients / result of EVERY step (no
        # ReLU record), enough to replay each Adam update in fp64 (tests/oracle_checks.py)
        self.probe_steps = ()
        self.probe_all_adam = False
        self.step_probes = {}

        self.clip_param = args.clip_param
        self.ppo_epoch = args.ppo_epoch
        self.num_mini_batch = args.num_mini_batch
        self.data_chunk_length = args.data_chunk_length
        self.value_loss_coef = args.value_loss_coef
        self.entropy_coef = args.entropy_coef
        self.max_grad_norm = args.max_grad_norm
        self.huber_delta = args.huber_delta
        self.gamma = args.gamma
        self._use_gae = args.use_gae
        self.gae_lambda = args.gae_lambda
        self._use_max_grad_norm = args.use_max_grad_norm
        # the PPO loss switches of config/algs/ippo.yaml (learners/ippo_learner.py:142-157,190-196,353-362) as kernel flags
        self._use_clipped_value_loss, self._use_huber_loss = args.use_clipped_value_loss, args.use_huber_loss
        self._use_value_active_masks, self._use_policy_active_masks = args.use_value_active_masks, args.use_policy_active_masks
        self._loss_flags = ((0 if args.use_huber_loss else L.PPO_MSE) | (0 if args.use_clipped_value_loss else L.PPO_NO_VCLIP)
                            | (0 if args.use_value_active_masks else L.PPO_VALUE_MEAN)
                            | (0 if args.use_policy_active_masks else L.PPO_POLICY_MEAN))

        self.actor_params = mac.parameters()
        self.critic_params = mac.critic_parameters()
        self.actor_optimizers = [FusedAdam([(mac.actor_arena, n)], lr=self.lr, eps=self.optim_eps,
                                           weight_decay=self.weight_decay) for n in range(self.n_agents)]
        self.critic_optimizers = [FusedAdam([(mac.critic_arena, n)], lr=self.critic_lr, eps=self.optim_eps,
                                            weight_decay=self.weight_decay) for n in range(self.n_agents)]
        self.last_train_info = None
        self.dp = None          # optional iplan_amd.parallel.DataParallel (gradient / statistic all-reduce)
        # data-parallel runs whose ranks hold DIFFERENT numbers of PPO rows (config 4: a global 256-episode buffer of which
        # the first 255 episodes are trained on -> the last rank drops one): the global row counts, set by the caller
        self.dp_global_rows = None      # PPO rows of all ranks (default: rows * world)
        self.dp_global_count = None     # stored (episode, step) entries of all ranks (default: bs * T * world)

    def _moment_snapshot(self):
        """diagnostics (``probe_last_step``): Adam's first / second moments of both arenas and the step counts as they are
        BEFORE the next optimiser step -- with ``last_step_params`` the complete state the last step started from"""
        from ..optim import _moments
        return tuple(tuple(t.clone() for t in _moments(arena)) + (opts[0]._steps,)
                     for arena, opts in ((self.mac.actor_arena, self.actor_optimizers), (self.mac.critic_arena, self.critic_optimizers)))

    def _probe_before(self, k, n_steps, out):
        """diagnostics: snapshot the state optimiser step ``k`` of ``n_steps`` starts from (gradients are in the arenas already)"""
        mac = self.mac
        last = self.probe_last_step and k == n_steps - 1
        full = last or k in self.probe_steps
        if not (full or self.probe_all_adam):
            return
        rec = dict(params=(mac.actor_arena.data.clone(), mac.critic_arena.data.clone()), moments=self._moment_snapshot())
        if full:
            rec["relu"] = (out["saved"][..., 0:64] > 0, out["saved"][..., 128:192] > 0)       # fc1 / fc2 ReLU branches taken
        self.step_probes[k] = rec
        if last:
            self.last_step_params, self.last_step_moments, self.last_step_relu = rec["params"], rec["moments"], rec["relu"]

    def _probe_after(self, k):
        rec = self.step_probes.get(k)
        if rec is not None and "post" not in rec:
            mac = self.mac
            rec["grads"] = (mac.actor_arena.grad.clone(), mac.critic_arena.grad.clone())      # clipped (the Adam launch writes them back)
            rec["post"] = (mac.actor_arena.data.clone(), mac.critic_arena.data.clone())

    def lr_decay(self, episode, episodes):
        for n in range(self.n_agents):
            update_linear_schedule(self.actor_optimizers[n], episode, episodes, self.lr)
            update_linear_schedule(self.critic_optimizers[n], episode, episodes, self.critic_lr)

    def insert_episode_batch(self, ep_batch):
        """learners/ippo_learner.py:96-126 -- one strided copy per field for all agents."""
        self.store.insert(ep_batch, min(self.batch_size_run, ep_batch.batch_size))

    # ------------------------------------------------------------------------------------------ train
    def _feature_spec(self, T, T_phys, last):
        a, d = self.args, self.store.data
        srcs = []
        for key, w in self.mac._widths():
            t = d[key]                                     # [bs, T1, nA, N, w]
            srcs.append((t, w, t.stride(2), t.stride(1)))
        return ops.AcFeatureSpec(a.max_vehicle_num, srcs, n_actions=a.n_actions if a.obs_last_action else 0,
                                 last_action=last, la_strides=(1, self.n_agents),
                                 n_id=self.n_agents if a.obs_agent_id else 0, T=T, T_phys=T_phys)

    def _last_action_index(self):
        """Hot index of the last-action block of every stored step ([bs, T1, nA] int32, -1 = an all-zero block): the
        reference feeds ``actions_onehot`` shifted by one step, ``actions_onehot[0]`` at t = 0
        (dcntrl_controller.py:105-108).  Steps a rollout never wrote (the runner breaks once every env has terminated,
        ippo_parallel_runner.py:212-214) keep ``actions`` = 0 but an all-zero ``actions_onehot`` row -- the OneHot
        preprocess only runs on updated slices -- so the index comes from the stored one-hot rows, not from ``actions``.
        Index bookkeeping only (the rows are exact one-hots or zeros)."""
        d = self.store.data
        oh = d.get("actions_onehot")
        if oh is None:
            idx = d["actions"][..., 0]
        else:
            idx = th.where(oh.sum(dim=-1) > 0, oh.argmax(dim=-1), th.full_like(oh[..., 0], -1, dtype=th.long))
        return th.cat([idx[:, :1], idx[:, :-1]], dim=1).to(th.int32).contiguous()

    def train(self, t_env, defer=False):
        """learners/ippo_learner.py:227-317.  ``defer=True``: enqueue everything on the current stream and return a
        ``finish()`` callable that does the single host read-back + logging (None when the buffer is not full)."""
        if not self.buffers[0].can_sample():
            return
        print("TRAINING IPPO")
        self.step_probes = {}
        if self.use_linear_lr_decay:
            self.lr_decay(t_env, self.t_max)
        a, d, nA = self.args, self.store.data, self.n_agents
        mac = self.mac
        bs, T = self.store.size, self.episode_limit
        T1 = T + 1
        dev = self.device
        f32 = dict(dtype=th.float32, device=dev)
        last = self._last_action_index()
        ha, hc = d["rnn_states_actors"], d["rnn_states_critics"]            # [bs, T1, nA, M]
        hs = (ha.stride(2), ha.stride(1))
        avail = d["avail_actions"]
        if avail.dtype != th.int32:
            avail = avail.to(th.int32)
        av_s = (avail.stride(2), avail.stride(1))
        actions = d["actions"]
        act_s = (actions.stride(2), actions.stride(1))
        n_act = a.n_actions

        # compute_returns (:344-365): critic on every stored step, GAE, advantage normalisation (:273-279)
        spec_all = self._feature_spec(T1, T1, last)
        # LayerNorm(F) statistics of every stored row: the features are fixed during train(), so this pass
        # computes them once and all 1 + 15 x 2 later passes re-use them
        ln_stats = th.empty(nA, bs * T1, 2, **f32)
        v_all = ops.ac_forward(None, mac.critic_arena, 1, spec_all, bs * T1, nA, h_critic=hc, h_strides=hs,
                               ksplit=1, want_h=False, ln_stats=ln_stats, ln_stats_mode=1, packed=mac.fc1_pack.get(spec_all))["values"]
        rw, tm = d["reward"], d["terminated"]
        pp = L.PpoPrepareArgs()
        pp.n_agents, pp.bs, pp.T = nA, bs, T
        pp.reward, pp.rw_s_net, pp.rw_s_ep, pp.rw_s_t = rw.data_ptr(), rw.stride(2), rw.stride(0), rw.stride(1)
        assert tm.dtype == th.uint8
        pp.terminated, pp.tm_s_net, pp.tm_s_ep, pp.tm_s_t = tm.data_ptr(), tm.stride(2), tm.stride(0), tm.stride(1)
        pp.values, pp.gamma, pp.lam = v_all.data_ptr(), self.gamma, self.gae_lambda
        returns, adv, mask, vpred = (th.empty(nA, bs * T, **f32) for _ in range(4))
        pp.returns, pp.adv, pp.mask, pp.value_preds = returns.data_ptr(), adv.data_ptr(), mask.data_ptr(), vpred.data_ptr()
        lib = L.get_lib()
        stream = L.current_stream(dev)
        pp.skip_norm = 0 if self.dp is None else 1
        pp.no_gae = 0 if self._use_gae else 1
        lib.call("iplan_ppo_prepare", pp, stream)
        if self.dp is not None:
            # data-parallel: advantage mean / unbiased std over ALL ranks' rows -- three launches of iplan_ppo_adv_norm
            # around two [nA]-float sum all-reduces (the host only moves the partial sums; no arithmetic here)
            an = L.AdvNormArgs()
            an.n_agents, an.n, an.row_stride = nA, bs * T, bs * T
            asum, asq = th.empty(nA, **f32), th.empty(nA, **f32)
            an.adv, an.sum, an.sqdev = adv.data_ptr(), asum.data_ptr(), asq.data_ptr()
            an.count = float(self.dp_global_count if self.dp_global_count is not None else bs * T * self.dp.world)
            for phase, red in ((0, asum), (1, asq), (2, None)):
                an.phase = phase
                lib.call("iplan_ppo_adv_norm", an, stream)
                if red is not None:
                    self.dp.all_reduce_sum(red)
        # generate_data (:368-424): the first batch_size * T rows
        rows = self.batch_size * T
        spec = self._feature_spec(T, T1, last)
        fwd_kw = dict(h_actor=ha, h_critic=hc, h_strides=hs, avail=avail, avail_strides=av_s, mode=2,
                      actions_in=actions, act_strides=act_s, n_actions=n_act, ksplit=1, want_h=False,
                      ln_stats=ln_stats, ln_stats_mode=2)
        # old_action_log_probs (:383): the actor's log-probs at the parameters train() starts from.  The full-batch split-bf16
        # path takes them from its own first epoch (same parameters, same kernels: ratio == 1 there, as in the reference,
        # which evaluates one network twice); the other paths need them up front.
        split = self.num_mini_batch == 1 and not os.environ.get("IPLAN_PPO_FC1_FP32")
        # The split-bf16 fc1 kernels are shaped for the full buffer (512 rows per workgroup, the whole K loop in one workgroup):
        # below IPLAN_PPO_SPLIT_MIN_ROWS rows per agent (a data-parallel rank's share of config 4: 2 880) their few workgroups
        # are one long latency chain each and the streaming fp32 form (16 rows per wave) is the faster one.
        if split and self.batch_size * T < int(os.environ.get("IPLAN_PPO_SPLIT_MIN_ROWS", "0")):
            split = False
        if split:
            # The split path keeps two packed copies of the normalised rows for the whole train() (2 x rows x Kpad x 4 bytes per
            # agent + the pre-activations: 2.3 GB at config 3, 18 GB for config 4 on one GPU -- sized for 288 GB of HBM3E).  On
            # a device where that estimate does not fit the budget (IPLAN_PPO_XHAT_BUDGET_GB, default a quarter of the device's
            # memory) the fp32 contraction, which streams the episode fields in place, runs instead: same results to fp32
            # round-off (tests/...::test_fc1_split_vs_fp32_*), ~1.8x the time.
            F_est = sum(w for _, w in mac._widths()) * a.max_vehicle_num + (a.n_actions if a.obs_last_action else 0) + (nA if a.obs_agent_id else 0)
            need = nA * self.batch_size * T * (2 * (F_est + 32) + 2 * 64) * 4
            budget = os.environ.get("IPLAN_PPO_XHAT_BUDGET_GB")
            if budget is not None:
                budget = float(budget) * 2 ** 30
            elif dev.type == "cuda":
                budget = th.cuda.get_device_properties(dev).total_memory / 4
            else:
                budget = float("inf")
            if need > budget:
                split = False
        if split:
            old_logp = th.zeros(nA, bs * T, **f32)
        else:
            old_logp = ops.ac_forward(mac.actor_arena, None, 0, spec, rows, nA, packed=mac.fc1_pack.get(spec), **fwd_kw)["logp"]
            if bs * T != rows:                               # pad to the [nA, bs*T] stride of adv / returns
                tmp = th.zeros(nA, bs * T, **f32)
                tmp[:, :rows] = old_logp
                old_logp = tmp

        if self.num_mini_batch > 1:
            fin = self._train_minibatches(t_env, rows, last, ln_stats, old_logp, adv, returns, vpred, mask)
            return fin if defer else (fin() and None)

        pl = L.PpoLossArgs()
        pl.n_agents, pl.rows, pl.row_stride = nA, rows, bs * T
        pl.old_logp, pl.adv, pl.value_preds = old_logp.data_ptr(), adv.data_ptr(), vpred.data_ptr()
        pl.returns, pl.mask = returns.data_ptr(), mask.data_ptr()
        pl.clip, pl.huber_delta, pl.value_loss_coef = self.clip_param, self.huber_delta, self.value_loss_coef
        g_logp, g_v = th.empty(nA, rows, **f32), th.empty(nA, rows, **f32)
        # the loss launch's rows in `parts` workgroups per agent (one workgroup walked 22 950 rows in 22 dependent rounds: 46.5 us
        # x 15 epochs): every workgroup needs the losses' denominator sum(mask) up front -- the mask does not change over the
        # epochs, so it is added up ONCE here -- and writes its range's share of the statistics, added up after the last epoch
        parts = max(1, min(rows, int(os.environ.get("IPLAN_PPO_LOSS_PARTS", min(64, rows // 1024)))))          # (the knob: tests, A/B)
        stats = th.zeros(self.ppo_epoch, nA, parts, 8, **f32)
        norms = th.zeros(self.ppo_epoch, 2, nA, **f32)
        pl.g_logp, pl.g_values = g_logp.data_ptr(), g_v.data_ptr()
        pl.flags = self._loss_flags
        pl.n_parts = parts
        max_norm = self.max_grad_norm if self._use_max_grad_norm else None
        n_rows = float(rows)
        msum = mask[:, :rows].sum(dim=1).contiguous() if (parts > 1 or self.dp is not None) else None     # (0 / 1 entries: exact in any order)
        if self.dp is not None:
            # the losses' denominators over all ranks: sum(mask) of the PPO rows and the row count (entropy mean)
            msum = self.dp.all_reduce_sum(msum)
            n_rows = float(self.dp_global_rows if self.dp_global_rows is not None else rows * self.dp.world)
            pl.row_count = n_rows                               # denominator of the unmasked (mean) loss forms
        if msum is not None:
            pl.mask_sum = msum.data_ptr()
        # the epochs re-evaluate the SAME rows: their normalised features are gathered once into the fragment-major arrays the
        # split-bf16 fc1 kernels stream (ops.ac_xhat_pack); IPLAN_PPO_FC1_FP32=1 keeps the fp32 contraction (A/B, diagnostics)
        xhat = ops.ac_xhat_pack(spec, rows, nA, ln_stats) if split else None
        for ep in range(self.ppo_epoch):
            out = ops.ac_forward(mac.actor_arena, mac.critic_arena, 2, spec, rows, nA, save=True, want_entropy=True, xhat=xhat,
                                 packed=None if xhat is not None else mac.fc1_pack.get(spec), **fwd_kw)   # (repacked after every Adam step)
            if split and ep == 0:
                old_logp[:, :rows].copy_(out["logp"])
            pl.logp, pl.entropy, pl.values = out["logp"].data_ptr(), out["entropy"].data_ptr(), out["values"].data_ptr()
            pl.stats = stats[ep].data_ptr()
            lib.call("iplan_ppo_loss", pl, stream)
            ops.ac_backward(out, mac.actor_arena, mac.critic_arena, g_logp=g_logp,
                            g_entropy=-self.entropy_coef / n_rows, g_values=g_v)
            if self.dp is not None:
                self.dp.all_reduce_grads(mac.actor_arena, mac.critic_arena)
            self._probe_before(ep, self.ppo_epoch, out)
            sq_a = step_all(self.actor_optimizers, max_norm)
            norms[ep, 0] = sq_a[:, 0]
            sq_c = step_all(self.critic_optimizers, max_norm)
            norms[ep, 1] = sq_c[:, 0]
            self._probe_after(ep)
        self.store.clear()

        # train_info (:305-310): averages over agents x epochs -- ONE host read-back
        st_d = stats.sum(dim=2).mean(dim=(0, 1))
        nr_d = norms.sqrt().mean(dim=(0, 2)) if max_norm is not None else th.zeros(2, device=dev)

        staged = AsyncHost(th.cat([st_d.reshape(-1), nr_d.reshape(-1)])) if defer else None     # read back whenever the caller likes

        def finish():
            if staged is not None:
                both = staged.get()
                st, nr = both[:st_d.numel()], both[st_d.numel():]
            else:
                st, nr = st_d.cpu(), nr_d.cpu()
            train_info = {"value_loss": float(st[1]), "policy_loss": float(st[0]), "dist_entropy": float(st[3]),
                          "actor_grad_norm": float(nr[0]), "critic_grad_norm": float(nr[1]), "ratio": float(st[2])}
            self.last_train_info = train_info
            if t_env - self.log_stats_t >= self.args.learner_log_interval:
                for k, v in train_info.items():
                    self.logger.log_stat(self.log_prefix + k, v, t_env)
            return train_info
        return finish if defer else (finish() and None)

    def _train_minibatches(self, t_env, rows, last, ln_stats, old_logp, adv, returns, vpred, mask):
        """num_mini_batch > 1 (generate_data, learners/ippo_learner.py:368-424): every epoch draws a fresh ``randperm`` of the
        rows per agent and steps the optimisers once per minibatch.  A minibatch is a set of INDEPENDENT rows (stored GRU
        states, one step each), so the epoch's rows are gathered once, in permuted order, into contiguous per-agent staging
        tensors (index plumbing: what the reference's ``obs[indices]`` does on the assembled 229 MB tensor, here on the raw
        fields) and the fused kernels run on row ranges of those.  The permutations are drawn in the reference's order
        (agent-major, one per epoch) from torch's global CPU generator."""
        a, d, nA, mac = self.args, self.store.data, self.n_agents, self.mac
        if self.dp is not None:
            return self._train_minibatches_dp(t_env, rows, last, ln_stats, old_logp, adv, returns, vpred, mask)
        dev = self.device
        f32 = dict(dtype=th.float32, device=dev)
        T, T1, nmb = self.episode_limit, self.episode_limit + 1, self.num_mini_batch
        mbs = rows // nmb
        perms = th.stack([th.stack([th.randperm(rows) for _ in range(self.ppo_epoch)]) for _ in range(nA)]).to(dev)   # [nA, epochs, rows]
        ag = th.arange(nA, device=dev)[:, None]
        lib = L.get_lib()
        stream = L.current_stream(dev)
        max_norm = self.max_grad_norm if self._use_max_grad_norm else None
        stats = th.zeros(self.ppo_epoch * nmb, nA, 8, **f32)
        norms = th.zeros(self.ppo_epoch * nmb, 2, nA, **f32)
        widths = mac._widths()
        avail_all = d["avail_actions"] if d["avail_actions"].dtype == th.int32 else d["avail_actions"].to(th.int32)
        for ep in range(self.ppo_epoch):
            idx = perms[:, ep, :mbs * nmb]                                   # [nA, n] rows of this epoch, minibatch-major
            e_i, t_i = idx // T, idx % T
            n = idx.shape[1]
            src = [(d[k][e_i, t_i, ag].reshape(nA, n, -1).contiguous(), w) for k, w in widths]            # [nA, n, N*w]
            la = last[e_i, t_i, ag].contiguous()                               # [nA, n] int32
            ha, hc = d["rnn_states_actors"][e_i, t_i, ag].contiguous(), d["rnn_states_critics"][e_i, t_i, ag].contiguous()
            av = avail_all[e_i, t_i, ag].contiguous()
            ac = d["actions"][e_i, t_i, ag].contiguous()                       # [nA, n, 1] int64
            lns = ln_stats.reshape(nA, -1, 2)[ag, e_i * T1 + t_i].contiguous()   # [nA, n, 2]
            g = lambda x: x[ag, idx].contiguous()                             # noqa: E731  -- [nA, bs*T] per-row quantities
            olp, adv_g, ret_g, vp_g, msk_g = g(old_logp), g(adv), g(returns), g(vpred), g(mask)
            for i in range(nmb):
                lo, hi = i * mbs, (i + 1) * mbs
                spec = ops.AcFeatureSpec(a.max_vehicle_num, [(t[:, lo:hi], w, t.stride(0), t.stride(1)) for t, w in src],
                                         n_actions=a.n_actions if a.obs_last_action else 0, last_action=la[:, lo:hi],
                                         la_strides=(la.stride(0), 1), n_id=nA if a.obs_agent_id else 0, T=1, T_phys=1)
                out = ops.ac_forward(mac.actor_arena, mac.critic_arena, 2, spec, mbs, nA, h_actor=ha[:, lo:hi], h_critic=hc[:, lo:hi],
                                     h_strides=(ha.stride(0), ha.stride(1)), avail=av[:, lo:hi], avail_strides=(av.stride(0), av.stride(1)),
                                     mode=2, actions_in=ac[:, lo:hi], act_strides=(ac.stride(0), ac.stride(1)), n_actions=a.n_actions,
                                     ksplit=1, want_h=False, ln_stats=lns[:, lo:hi], ln_stats_mode=2, save=True, want_entropy=True,
                                     packed=mac.fc1_pack.get(spec))
                pl = L.PpoLossArgs()
                pl.n_agents, pl.rows, pl.row_stride = nA, mbs, n
                pl.old_logp, pl.adv, pl.value_preds = olp[:, lo:hi].data_ptr(), adv_g[:, lo:hi].data_ptr(), vp_g[:, lo:hi].data_ptr()
                pl.returns, pl.mask = ret_g[:, lo:hi].data_ptr(), msk_g[:, lo:hi].data_ptr()
                pl.clip, pl.huber_delta, pl.value_loss_coef = self.clip_param, self.huber_delta, self.value_loss_coef
                g_logp, g_v = th.empty(nA, mbs, **f32), th.empty(nA, mbs, **f32)
                pl.g_logp, pl.g_values = g_logp.data_ptr(), g_v.data_ptr()
                pl.flags = self._loss_flags
                pl.logp, pl.entropy, pl.values = out["logp"].data_ptr(), out["entropy"].data_ptr(), out["values"].data_ptr()
                k = ep * nmb + i
                pl.stats = stats[k].data_ptr()
                lib.call("iplan_ppo_loss", pl, stream)
                ops.ac_backward(out, mac.actor_arena, mac.critic_arena, g_logp=g_logp, g_entropy=-self.entropy_coef / float(mbs), g_values=g_v)
                self._probe_before(k, self.ppo_epoch * nmb, out)
                norms[k, 0] = step_all(self.actor_optimizers, max_norm)[:, 0]
                norms[k, 1] = step_all(self.critic_optimizers, max_norm)[:, 0]
                self._probe_after(k)
        self.store.clear()
        st_d = stats.mean(dim=(0, 1))
        nr_d = norms.sqrt().mean(dim=(0, 2)) if max_norm is not None else th.zeros(2, device=dev)

        staged = AsyncHost(th.cat([st_d.reshape(-1), nr_d.reshape(-1)]))

        def finish():
            both = staged.get()
            st, nr = both[:st_d.numel()], both[st_d.numel():]
            train_info = {"value_loss": float(st[1]), "policy_loss": float(st[0]), "dist_entropy": float(st[3]),
                          "actor_grad_norm": float(nr[0]), "critic_grad_norm": float(nr[1]), "ratio": float(st[2])}
            self.last_train_info = train_info
            if t_env - self.log_stats_t >= self.args.learner_log_interval:
                for kk, v in train_info.items():
                    self.logger.log_stat(self.log_prefix + kk, v, t_env)
            return train_info
        return finish

    def _train_minibatches_dp(self, t_env, rows, last, ln_stats, old_logp, adv, returns, vpred, mask):
        """num_mini_batch > 1 under data parallelism: the minibatches are those of ONE process holding the
        union of the ranks' rows.  Every rank draws the reference's permutations of the GLOBAL row range (agent-major, one per
        epoch, torch's global CPU generator -- exactly what the single process would draw), rank 0's are broadcast; minibatch i of
        an epoch is the global rows ``perm[i * mbs : (i + 1) * mbs]``, and every rank trains on the ones that fall into its own
        row range.  Their number differs per agent and rank, while one fused launch covers all agents with one row count: each
        agent's selection is padded to the launch's row count with rows of ZERO weight (mask 0 -> no policy / value term,
        per-row entropy weight 0), the masked-mean denominators and the entropy count are the global minibatch's (one
        all-reduce of all mask sums up front), and the gradient arenas are summed over the ranks before every step -- so an
        N-rank step equals the union's step.  The unmasked-mean loss forms (use_*_active_masks off) weight every row of a
        launch alike and cannot ignore padding: not available in this combination."""
        a, d, nA, mac, dp = self.args, self.store.data, self.n_agents, self.mac, self.dp
        if not (self._use_value_active_masks and self._use_policy_active_masks):
            raise NotImplementedError("data-parallel num_mini_batch > 1 needs use_value_active_masks and use_policy_active_masks")
        dev = self.device
        f32 = dict(dtype=th.float32, device=dev)
        T, T1, nmb = self.episode_limit, self.episode_limit + 1, self.num_mini_batch
        # this rank's place in the union's row order: ranks in order, each with its own rows
        counts = th.zeros(dp.world, dtype=th.int64, device=dev)
        counts[dp.rank] = rows
        counts = dp.all_reduce_sum(counts).cpu()                              # (int64: exact at any row count)
        G = int(counts.sum())
        if self.dp_global_rows is not None:
            assert G == int(self.dp_global_rows), (G, self.dp_global_rows)
        off = int(counts[:dp.rank].sum())
        mbs = G // nmb
        # EVERY rank draws (the ranks' CPU generators then stay in step for whatever is drawn from them next: the gumbel seed,
        # the dropout seed, the prediction samples); rank 0's draw is the one used -- identical to the others' when the ranks
        # were seeded alike, and the broadcast makes it so when they were not
        perms = th.stack([th.stack([th.randperm(G) for _ in range(self.ppo_epoch)]) for _ in range(nA)])
        perms = dp.broadcast_tensor(perms.to(dev))
        steps = self.ppo_epoch * nmb
        mb = perms[:, :, :mbs * nmb].reshape(nA, steps, mbs)                  # global rows of every (agent, step)
        mine = (mb >= off) & (mb < off + rows)
        n_loc = mine.sum(-1)                                                  # [nA, steps]
        n_pad = n_loc.max(0).values.clamp_min(1).cpu().tolist()               # launch row count per step
        # local selections, own rows first (stable), padded with this rank's row 0 at weight 0
        order = th.argsort((~mine).to(th.int8), dim=-1, stable=True)
        loc = (th.gather(mb, -1, order) - off).clamp_(0, rows - 1)            # [nA, steps, mbs]; beyond n_loc: padding
        w_all = (th.arange(mbs, device=dev)[None, None, :] < n_loc[..., None]).to(th.float32)
        ag = th.arange(nA, device=dev)[:, None]
        # the masked-mean denominators of every global minibatch in ONE all-reduce
        msk_loc = th.gather(mask[:, None, :].expand(nA, steps, mask.shape[1]), -1, loc) * w_all
        msum_all = dp.all_reduce_sum(msk_loc.sum(-1).t().contiguous())       # [steps, nA]
        lib = L.get_lib()
        stream = L.current_stream(dev)
        max_norm = self.max_grad_norm if self._use_max_grad_norm else None
        stats = th.zeros(steps, nA, 8, **f32)
        norms = th.zeros(steps, 2, nA, **f32)
        widths = mac._widths()
        avail_all = d["avail_actions"] if d["avail_actions"].dtype == th.int32 else d["avail_actions"].to(th.int32)
        for k in range(steps):
            n = int(n_pad[k])
            idx, w = loc[:, k, :n], w_all[:, k, :n].contiguous()              # [nA, n]
            e_i, t_i = idx // T, idx % T
            src = [(d[key][e_i, t_i, ag].reshape(nA, n, -1).contiguous(), wd) for key, wd in widths]
            la = last[e_i, t_i, ag].contiguous()
            ha, hc = d["rnn_states_actors"][e_i, t_i, ag].contiguous(), d["rnn_states_critics"][e_i, t_i, ag].contiguous()
            av = avail_all[e_i, t_i, ag].contiguous()
            ac = d["actions"][e_i, t_i, ag].contiguous()
            lns = ln_stats.reshape(nA, -1, 2)[ag, e_i * T1 + t_i].contiguous()
            g = lambda x: x[ag, idx].contiguous()                             # noqa: E731
            olp, adv_g, ret_g, vp_g = g(old_logp), g(adv), g(returns), g(vpred)
            msk_g = (g(mask) * w).contiguous()
            spec = ops.AcFeatureSpec(a.max_vehicle_num, [(t, wd, t.stride(0), t.stride(1)) for t, wd in src],
                                     n_actions=a.n_actions if a.obs_last_action else 0, last_action=la,
                                     la_strides=(la.stride(0), 1), n_id=nA if a.obs_agent_id else 0, T=1, T_phys=1)
            out = ops.ac_forward(mac.actor_arena, mac.critic_arena, 2, spec, n, nA, h_actor=ha, h_critic=hc,
                                 h_strides=(ha.stride(0), ha.stride(1)), avail=av, avail_strides=(av.stride(0), av.stride(1)),
                                 mode=2, actions_in=ac, act_strides=(ac.stride(0), ac.stride(1)), n_actions=a.n_actions,
                                 ksplit=1, want_h=False, ln_stats=lns, ln_stats_mode=2, save=True, want_entropy=True,
                                 packed=mac.fc1_pack.get(spec))
            pl = L.PpoLossArgs()
            pl.n_agents, pl.rows, pl.row_stride = nA, n, n
            pl.old_logp, pl.adv, pl.value_preds = olp.data_ptr(), adv_g.data_ptr(), vp_g.data_ptr()
            pl.returns, pl.mask = ret_g.data_ptr(), msk_g.data_ptr()
            pl.clip, pl.huber_delta, pl.value_loss_coef = self.clip_param, self.huber_delta, self.value_loss_coef
            g_logp, g_v = th.empty(nA, n, **f32), th.empty(nA, n, **f32)
            pl.g_logp, pl.g_values = g_logp.data_ptr(), g_v.data_ptr()
            pl.flags = self._loss_flags
            pl.logp, pl.entropy, pl.values = out["logp"].data_ptr(), out["entropy"].data_ptr(), out["values"].data_ptr()
            pl.mask_sum, pl.row_count = msum_all[k].data_ptr(), float(mbs)
            pl.stats = stats[k].data_ptr()
            lib.call("iplan_ppo_loss", pl, stream)
            ops.ac_backward(out, mac.actor_arena, mac.critic_arena, g_logp=g_logp, g_entropy=(w * (-self.entropy_coef / float(mbs))).contiguous(),
                            g_values=g_v)
            dp.all_reduce_grads(mac.actor_arena, mac.critic_arena)
            self._probe_before(k, steps, out)
            norms[k, 0] = step_all(self.actor_optimizers, max_norm)[:, 0]
            norms[k, 1] = step_all(self.critic_optimizers, max_norm)[:, 0]
            self._probe_after(k)
        self.store.clear()
        st_d = stats.mean(dim=(0, 1))
        nr_d = norms.sqrt().mean(dim=(0, 2)) if max_norm is not None else th.zeros(2, device=dev)
        staged = AsyncHost(th.cat([st_d.reshape(-1), nr_d.reshape(-1)]))

        def finish():
            both = staged.get()
            st, nr = both[:st_d.numel()], both[st_d.numel():]
            # (policy / value losses are this rank's share of the global masked means; ratio and entropy include the padding rows)
            train_info = {"value_loss": float(st[1]), "policy_loss": float(st[0]), "dist_entropy": float(st[3]),
                          "actor_grad_norm": float(nr[0]), "critic_grad_norm": float(nr[1]), "ratio": float(st[2])}
            self.last_train_info = train_info
            if t_env - self.log_stats_t >= self.args.learner_log_interval:
                for kk, v in train_info.items():
                    self.logger.log_stat(self.log_prefix + kk, v, t_env)
            return train_info
        return finish

    # ------------------------------------------------------------------ reference-shaped single-agent methods
    # The fused train() above is the production path.  The methods below keep the reference's per-agent,
    # per-minibatch surface (learners/ippo_learner.py:128-225, 344-424) for callers that drive PPO themselves;
    # they run the same kernels through the module-level autograd Functions.
    def compute_returns(self, agent_id, obs_all, rewards, terminated, rnn_state_critic_all):
        """learners/ippo_learner.py:344-365.  obs_all [bs,T+1,F] (assembled), rewards [bs,T,1], terminated =
        the reference's ``terminated_masks`` (1 - terminated) [bs,T+1,1] -> returns [bs,T,1]."""
        bs, T = rewards.shape[0], rewards.shape[1]
        with th.no_grad():
            v_all = self.mac.get_value_ippo(agent_id, obs_all, rnn_state_critic_all).reshape(1, bs * (T + 1)).contiguous()
        rw = rewards.to(**self.tpdv).reshape(bs, T).contiguous()
        tm = (1 - terminated.to(self.device).reshape(bs, T + 1).to(th.float32)).to(th.uint8).contiguous()
        pp = L.PpoPrepareArgs()
        pp.n_agents, pp.bs, pp.T = 1, bs, T
        pp.reward, pp.rw_s_net, pp.rw_s_ep, pp.rw_s_t = rw.data_ptr(), 0, T, 1
        pp.terminated, pp.tm_s_net, pp.tm_s_ep, pp.tm_s_t = tm.data_ptr(), 0, T + 1, 1
        pp.values, pp.gamma, pp.lam = v_all.data_ptr(), self.gamma, self.gae_lambda
        outs = [th.empty(1, bs * T, **self.tpdv) for _ in range(4)]
        pp.returns, pp.adv, pp.mask, pp.value_preds = (o.data_ptr() for o in outs)
        pp.no_gae = 0 if self._use_gae else 1
        L.get_lib().call("iplan_ppo_prepare", pp, L.current_stream(self.device))
        return outs[0].reshape(bs, T, 1)

    def _losses(self, logp, entropy_rows, values, old_logp, adv, value_preds, returns, mask):
        """One iplan_ppo_loss launch for a single agent: (stats [8], dLoss/dlogp [R], d(value_loss)/dvalue [R])."""
        R = values.numel()
        f = lambda t: t.detach().to(**self.tpdv).reshape(1, R).contiguous()  # noqa: E731
        pl = L.PpoLossArgs()
        pl.n_agents, pl.rows, pl.row_stride = 1, R, R
        ts = [f(t) for t in (logp, entropy_rows, values, old_logp, adv, value_preds, returns, mask)]
        (pl.logp, pl.entropy, pl.values, pl.old_logp, pl.adv, pl.value_preds, pl.returns, pl.mask) = (t.data_ptr() for t in ts)
        pl.clip, pl.huber_delta, pl.value_loss_coef = self.clip_param, self.huber_delta, 1.0
        g_lp, g_v, stats = th.empty(1, R, **self.tpdv), th.empty(1, R, **self.tpdv), th.zeros(1, 8, **self.tpdv)
        pl.g_logp, pl.g_values, pl.stats = g_lp.data_ptr(), g_v.data_ptr(), stats.data_ptr()
        pl.flags = self._loss_flags
        L.get_lib().call("iplan_ppo_loss", pl, L.current_stream(self.device))
        return stats[0], g_lp[0], g_v[0]

    def cal_value_loss(self, values, value_preds_batch, return_batch, terminated_batch):
        """learners/ippo_learner.py:128-159 (clipped one-sided Huber, masked mean) -- differentiable w.r.t. ``values``."""
        return _ValueLoss.apply(self, values, value_preds_batch, return_batch, terminated_batch)

    def generate_data(self, obs, rnn_states_actor, rnn_states_critic, actions, returns, terminated, action_log_probs,
                      advantages, available_actions, value_preds, num_mini_batch=None, mini_batch_size=None):
        """learners/ippo_learner.py:368-424: shuffled minibatches over the first batch_size * episode_limit rows
        (index plumbing only)."""
        n = self.batch_size * self.episode_limit
        if mini_batch_size is None:
            mini_batch_size = n // num_mini_batch
        rand = th.randperm(n)
        flat = lambda t: t.reshape(-1, *t.shape[2:])  # noqa: E731
        obs, ha, hc = flat(obs), flat(rnn_states_actor), flat(rnn_states_critic)
        actions = actions.reshape(-1, actions.shape[-1])
        if available_actions is not None:
            available_actions = available_actions[:-1].reshape(-1, available_actions.shape[-1])
        vp, rt, tm = value_preds.reshape(-1, 1), returns.reshape(-1, 1), terminated.reshape(-1, 1)
        lp, adv = action_log_probs.reshape(-1, action_log_probs.shape[-1]), advantages.reshape(-1, 1)
        for i in range(num_mini_batch):
            idx = rand[i * mini_batch_size:(i + 1) * mini_batch_size].to(self.device)
            yield (obs[idx], ha[idx], hc[idx], actions[idx], vp[idx], rt[idx], tm[idx], lp[idx], adv[idx],
                   None if available_actions is None else available_actions[idx])

    def ppo_update(self, agent_id, obs_batch, rnn_states_actor_batch, rnn_states_critic_batch, actions_batch,
                   value_preds_batch, return_batch, terminated_batch, old_action_log_probs_batch, adv_targ,
                   available_actions_batch, update_actor=True):
        """learners/ippo_learner.py:161-225 for one agent and one minibatch."""
        logp, ent = self.mac.eval_action_ippo(agent_id, obs_batch, actions_batch, available_actions_batch, rnn_states_actor_batch)
        values = self.mac.get_value_ippo(agent_id, obs_batch, rnn_states_critic_batch)
        policy_loss, ratio_mean = _PolicyLoss.apply(self, logp, old_action_log_probs_batch, adv_targ, terminated_batch)
        max_norm = self.max_grad_norm if self._use_max_grad_norm else None
        self.actor_optimizers[agent_id].zero_grad()
        if update_actor:
            (policy_loss - ent * self.entropy_coef).backward()
        self.actor_optimizers[agent_id].step(max_norm=max_norm)
        actor_grad_norm = self.actor_optimizers[agent_id].grad_norms()[0]
        value_loss = self.cal_value_loss(values, value_preds_batch, return_batch, terminated_batch)
        self.critic_optimizers[agent_id].zero_grad()
        (value_loss * self.value_loss_coef).backward()
        self.critic_optimizers[agent_id].step(max_norm=max_norm)
        critic_grad_norm = self.critic_optimizers[agent_id].grad_norms()[0]
        return value_loss, critic_grad_norm, policy_loss, ent, actor_grad_norm, ratio_mean

    # ------------------------------------------------------------------------------------------ misc
    def cuda(self):
        self.mac.cuda()

    def save_models(self, path):
        self.mac.save_models(path)
        for i in range(self.n_agents):
            th.save(self.actor_optimizers[i].state_dict(), "{}/actor_{}_opt.th".format(path, i))
            th.save(self.critic_optimizers[i].state_dict(), "{}/critic_{}_opt.th".format(path, i))

    def load_models(self, paths, load_optimisers=False):
        self.mac.load_models(paths)
        if load_optimisers:
            if len(paths) == 1:
                paths = [copy.copy(paths[0]) for _ in range(self.n_agents)]
            for i in range(self.n_agents):
                self.actor_optimizers[i].load_state_dict(th.load("{}/actor_{}_opt.th".format(paths[i], i), map_location="cpu"))
                self.critic_optimizers[i].load_state_dict(th.load("{}/critic_{}_opt.th".format(paths[i], i), map_location="cpu"))


class _ValueLoss(th.autograd.Function):
    @staticmethod
    def forward(ctx, learner, values, value_preds, returns, mask):
        z = th.zeros_like(values)
        stats, _, g_v = learner._losses(z, z, values, z, z, value_preds, returns, mask)
        ctx.save_for_backward(g_v.reshape(values.shape))
        return stats[1].clone()

    @staticmethod
    def backward(ctx, g):
        (g_v,) = ctx.saved_tensors
        return None, g * g_v, None, None, None


class _PolicyLoss(th.autograd.Function):
    """(policy_loss, mean importance weight) of learners/ippo_learner.py:185-197."""

    @staticmethod
    def forward(ctx, learner, logp, old_logp, adv, mask):
        z = th.zeros_like(logp)
        stats, g_lp, _ = learner._losses(logp, z, z, old_logp, adv, z, z, mask)
        ctx.save_for_backward(g_lp.reshape(logp.shape))
        ctx.mark_non_differentiable(stats)
        return stats[0].clone(), stats[2].clone()

    @staticmethod
    def backward(ctx, g, _g_ratio):
        (g_lp,) = ctx.saved_tensors
        return None, g * g_lp, None, None, None
