"""DcntrlMAC -- decentralised multi-agent controller without parameter sharing (mirror of
controllers/dcntrl_controller.py:9-232).

Same constructor and method contracts as the reference.  Differences are internal:
  * the n_agents private R_Actor / R_Critic modules keep their own ``state_dict`` (checkpoints
    interchange with the reference) but their weights live in two stacked arenas, so
    ``select_actions_ippo`` is ONE fused launch for all agents, actors and critics, instead of
    2 * n_agents module calls;
  * the observation assembly of ``_build_inputs`` happens inside the kernel -- the [E, nA, F]
    input tensor is never materialised.  ``_build_inputs`` / ``_build_inputs_ippo`` remain available
    (plain tensor plumbing) for callers that want the assembled tensor.
"""
import copy

import torch as th

from .. import ops
from ..arena import ParamArena
from ..modules.agents.ippo_actor import R_Actor
from ..modules.critics.ippo_critic import R_Critic


class DcntrlMAC:
    def __init__(self, scheme, groups, args):
        self.n_agents = args.n_agents
        self.args = args
        self.device = th.device("cuda" if args.use_cuda else "cpu")
        input_shape = self._get_input_shape(scheme)
        self.input_shape = input_shape
        self._build_agents(input_shape)
        self._build_critics(input_shape)
        self.actor_arena = ParamArena(self.agents, self.device)
        self.critic_arena = ParamArena(self.critics, self.device)
        # actor + critic gradients in ONE buffer: a data-parallel PPO step exchanges them as one collective (15 per train()
        # instead of 30; the exchange is latency-bound)
        ParamArena.colocate_grads([self.actor_arena, self.critic_arena])
        for i in range(self.n_agents):
            self.agents[i].attach(self.actor_arena, i)
            self.critics[i].attach(self.critic_arena, i)
        self.fc1_pack = ops.Fc1Pack(self.actor_arena, self.critic_arena)     # fragment-major fc1 operands, repacked when stale
        self.agent_output_type = args.agent_output_type
        self.hidden_states = None
        self.input_scheme = scheme

    # ------------------------------------------------------------------------------ IPPO
    def _widths(self):
        a = self.args
        src = [("history", a.obs_shape_single)]
        if a.GAT_enable:
            src.append(("attention_latent", a.attention_dim))
        if a.Behavior_enable:
            src.append(("behavior_latent", a.latent_dim))
        return src

    def _dev(self, t, dtype=None):
        t = t.to(self.device) if t.device != self.device else t
        return t if dtype is None or t.dtype == dtype else t.to(dtype)

    def select_actions_ippo(self, ep_batch, t_ep, test_mode=False, q_noise=None, as_numpy=True, write_back=False, phase_clocks=None, launch=True):
        """One fused launch: features gathered in place from ``ep_batch`` at ``t_ep``, all agents,
        actor + critic (controllers/dcntrl_controller.py:27-58).  Returns the reference's 5-tuple
        (values [E,nA], actions [E,nA], list of nA logp [E,1], rnn_states_actors [1,E,nA,M],
        rnn_states_critics [1,E,nA,M]); numpy for the array items unless ``as_numpy=False``.
        ``write_back=True`` (device-resident rollouts): the kernel additionally writes the actions, their
        one-hot and the new GRU states straight into ``ep_batch`` (actions / actions_onehot at ``t_ep``, rnn
        states at ``t_ep + 1``), i.e. the ``EpisodeBatch.update`` of ippo_parallel_runner.py:260-266.
        ``launch=False`` (with ``write_back``): nothing is enqueued; returns the prepared launch for
        ``Prediction_policy.GAT_latent_update(..., fuse_ac=...)`` of step ``t_ep - 1``, whose launch then carries this action
        selection behind the latent updates it reads (ops.gat_forward)."""
        a = self.args
        nA, N, M = self.n_agents, a.max_vehicle_num, a.rnn_hidden_dim
        E = ep_batch.batch_size
        sources = []
        for key, w in self._widths():
            full = self._dev(ep_batch[key], th.float32)                  # [E, T1, nA, N, w]
            view = full[:, t_ep]
            sources.append((view, w, view.stride(1), view.stride(0)))
        last = None
        la_strides = (0, 0)
        if a.obs_last_action and t_ep > 0:
            last = self._dev(ep_batch["actions"])[:, t_ep - 1, :, 0]             # [E, nA] int64 view, read in place
            la_strides = (last.stride(1), last.stride(0))
        spec = ops.AcFeatureSpec(N, sources, n_actions=a.n_actions if a.obs_last_action else 0,
                                 last_action=last, la_strides=la_strides,
                                 n_id=nA if a.obs_agent_id else 0, T=E, T_phys=E)
        assert spec.F == self.input_shape, (spec.F, self.input_shape)
        avail = self._dev(ep_batch["avail_actions"])[:, t_ep]            # [E, nA, n_act] int32 view
        if avail.dtype != th.int32:
            avail = avail.to(th.int32)
        ha = self._dev(ep_batch["rnn_states_actors"], th.float32)[:, t_ep]       # [E, nA, M]
        hc = self._dev(ep_batch["rnn_states_critics"], th.float32)[:, t_ep]
        assert ha.stride() == hc.stride()
        if not test_mode and q_noise is None:
            q_noise = th.empty(nA, E, a.n_actions, dtype=th.float32, device=self.device).exponential_()
        wb = {}
        if write_back:
            ha_n = ep_batch["rnn_states_actors"][:, t_ep + 1]             # [E, nA, M] views
            hc_n = ep_batch["rnn_states_critics"][:, t_ep + 1]
            act_n = ep_batch["actions"][:, t_ep, :, 0]
            oh_n = ep_batch["actions_onehot"][:, t_ep]
            assert ha_n.stride() == hc_n.stride()
            wb = dict(h_out=(ha_n, hc_n, (ha_n.stride(1), ha_n.stride(0))),
                      actions_out=(act_n, (act_n.stride(1), act_n.stride(0))),
                      onehot_out=(oh_n, (oh_n.stride(1), oh_n.stride(0))))
        o = ops.ac_forward(self.actor_arena, self.critic_arena, 2, spec, E, nA, h_actor=ha, h_critic=hc,
                           h_strides=(ha.stride(1), ha.stride(0)), avail=avail,
                           avail_strides=(avail.stride(1), avail.stride(0)),
                           mode=0 if test_mode else 1, q_noise=q_noise, n_actions=a.n_actions, phase_clocks=phase_clocks,
                           packed=self.fc1_pack.get(spec, fold=E <= 512), launch=launch, **wb)
        if not launch:
            assert write_back
            return o
        values = o["values"].t()                                          # [E, nA]
        logps = [o["logp"][i].reshape(E, 1) for i in range(nA)]
        if write_back:
            actions, ha_new, hc_new = act_n, ha_n.unsqueeze(0), hc_n.unsqueeze(0)
        else:
            actions = o["actions"].t()
            ha_new = o["h_actor"].permute(1, 0, 2).unsqueeze(0)           # [1, E, nA, M]
            hc_new = o["h_critic"].permute(1, 0, 2).unsqueeze(0)
        if as_numpy:
            return (values.cpu().numpy(), actions.cpu().numpy(), logps,
                    ha_new.cpu().numpy(), hc_new.cpu().numpy())
        return values, actions, logps, ha_new, hc_new

    def get_value_ippo(self, agent_id, obs, rnn_states_critic):
        """controllers/dcntrl_controller.py:61-68."""
        obs_in = obs.reshape(-1, 1, obs.shape[-1])
        hidden_in = rnn_states_critic.reshape(self.args.recurrent_N, -1, self.args.rnn_hidden_dim)
        value, _ = self.critics[agent_id](obs_in, hidden_in)
        return value.reshape(*obs.shape[:-1], 1)

    def eval_action_ippo(self, agent_id, obs, action, available_actions, rnn_states_actor):
        """controllers/dcntrl_controller.py:70-85."""
        obs_in = obs.reshape(-1, 1, obs.shape[-1])
        hidden_in = rnn_states_actor.reshape(self.args.recurrent_N, -1, self.args.rnn_hidden_dim)
        action_in = action.reshape(-1, 1, 1)
        avail_in = available_actions.reshape(-1, 1, available_actions.shape[-1])
        logp, ent = self.agents[agent_id].evaluate_actions(obs_in, hidden_in, action_in, avail_in)
        return logp.reshape(*obs.shape[:-1], 1), ent

    def _build_inputs_ippo(self, agent_id, batch, action_onehot, discr_signal=None):
        """Assembled [bs, T, F] tensor for callers that want it (dcntrl_controller.py:87-115);
        pure concatenation, no arithmetic.  The fused learner path does not use it."""
        bs, num_ts = batch["history"].shape[:2]
        states = [batch["history"]]
        if self.args.GAT_enable:
            states.append(batch["attention_latent"])
        if self.args.Behavior_enable:
            states.append(batch["behavior_latent"])
        inputs = [th.cat(states, dim=-1).reshape(bs, num_ts, -1)]
        if self.args.obs_last_action:
            inputs.append(th.cat([action_onehot[:, 0].unsqueeze(1), action_onehot[:, :-1]], dim=1))
        if self.args.obs_agent_id:
            onehot = th.zeros((bs, num_ts, self.n_agents), device=inputs[0].device)
            onehot[:, :, agent_id] = 1
            inputs.append(onehot)
        return th.cat(inputs, dim=-1)

    def _build_inputs(self, batch, t):
        """dcntrl_controller.py:187-213 (assembled [bs, nA, F]; tensor plumbing only)."""
        bs = batch.batch_size
        states = [batch["history"][:, t]]
        if self.args.GAT_enable:
            states.append(batch["attention_latent"][:, t])
        if self.args.Behavior_enable:
            states.append(batch["behavior_latent"][:, t])
        inputs = [th.cat(states, dim=-1)]
        if self.args.obs_last_action:
            inputs.append(th.zeros_like(batch["actions_onehot"][:, t]) if t == 0 else batch["actions_onehot"][:, t - 1])
        if self.args.obs_agent_id:
            inputs.append(th.eye(self.n_agents, device=inputs[0].device).unsqueeze(0).expand(bs, -1, -1))
        return th.cat([x.reshape(bs, self.n_agents, -1) for x in inputs], dim=2)

    # ------------------------------------------------------------------------------ bookkeeping
    def init_hidden(self, batch_size):
        self.hidden_states = None

    def parameters(self):
        return [list(agent.parameters()) for agent in self.agents]

    def critic_parameters(self):
        return [list(critic.parameters()) for critic in self.critics]

    def load_state(self, other_mac):
        for i, agent in enumerate(self.agents):
            agent.load_state_dict(other_mac.agents[i].state_dict())

    def cuda(self):
        pass                                  # arenas are created on args.device already

    def set_train_mode(self):
        for m in self.agents + self.critics:
            m.train()

    def set_eval_mode(self):
        for m in self.agents + self.critics:
            m.eval()

    def save_models(self, path):
        for i, agent in enumerate(self.agents):
            th.save(agent.state_dict(), f"{path}/agent_{i}.th")
        for i, critic in enumerate(self.critics):
            th.save(critic.state_dict(), f"{path}/critic_{i}.th")

    def load_models(self, paths):
        if len(paths) == 1:
            paths = [copy.copy(paths[0]) for _ in range(self.n_agents)]
        for i, agent in enumerate(self.agents):
            agent.load_state_dict(th.load(f"{paths[i]}/agent_{i}.th", map_location="cpu"))
        for i, critic in enumerate(self.critics):
            critic.load_state_dict(th.load(f"{paths[i]}/critic_{i}.th", map_location="cpu"))

    def _build_agents(self, input_shape):
        self.agents = [R_Actor(input_shape, self.args) for _ in range(self.n_agents)]

    def _build_critics(self, input_shape):
        self.critics = []
        if self.args.critic is not None:
            self.critics = [R_Critic(input_shape, self.args) for _ in range(self.n_agents)]

    def _get_input_shape(self, scheme):
        h = scheme["history"]["vshape"]
        shape = h[0] * h[1]
        if self.args.GAT_enable:
            s = scheme["attention_latent"]["vshape"]
            shape += s[0] * s[1]
        if self.args.Behavior_enable:
            s = scheme["behavior_latent"]["vshape"]
            shape += s[0] * s[1]
        if self.args.obs_last_action:
            shape += scheme["actions_onehot"]["vshape"][0]
        if self.args.obs_agent_id:
            shape += self.n_agents
        return shape
