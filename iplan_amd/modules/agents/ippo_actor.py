"""R_Actor -- recurrent policy network (mirror of modules/agents/ippo_actor.py:10-102).

Same constructor signature, ``forward`` / ``evaluate_actions`` contracts and ``state_dict`` keys as
the reference; the arithmetic is the fused HIP kernel ``iplan_ac_fwd`` (+ ``iplan_ac_bwd``).
"""
import torch
import torch.nn as nn

from ... import ops
from ...arena import ParamArena
from ...utils.mappo_utils.blocks import ACTLayer, MLPBase, RNNLayer
from ...utils.mappo_utils.util import check


class _FusedNet(nn.Module):
    """Shared plumbing of R_Actor / R_Critic: arena attachment + pre-assembled-input launches."""

    def __init__(self, args):
        super().__init__()
        self.device = torch.device("cuda" if args.use_cuda else "cpu")
        self.rnn_hidden_dim = args.rnn_hidden_dim
        self._n_actions = args.n_actions
        self._arena = None
        self._net = 0
        self.act_tanh = not args.use_ReLU      # mlp.py:10 [nn.Tanh(), nn.ReLU()][use_ReLU]; ParamArena hands it to the kernels
        if args.rnn_hidden_dim != 64 or args.mlp_hidden_dim != 64 or args.layer_N != 1 \
                or not args.use_feature_normalization or not args.use_recurrent_policy or args.recurrent_N != 1:
            raise NotImplementedError("iplan_amd builds the actor/critic kernel for the reference's shipped "
                                      "configuration (hidden 64, layer_N 1, ReLU or tanh, feature norm, 1 GRU layer)")

    def attach(self, arena, net):
        self._arena, self._net = arena, net

    def _own_arena(self, device):
        if self._arena is None:
            self._arena = ParamArena([self], device)
            self._net = 0
        return self._arena

    def _single(self):
        from ...nova.GAT_Net import _SingleNetView
        arena = self._own_arena(self.device)
        return arena if arena.n_nets == 1 else _SingleNetView(arena, self._net)

    def cuda(self, device=None):           # parameters already live in the device arena
        return self

    def _spec(self, obs):
        """obs [R, 1, F] (already assembled) -> feature descriptor with one 'entity' of width F."""
        x = check(obs).to(dtype=torch.float32, device=self.device).reshape(-1, obs.shape[-1]).contiguous()
        R, F = x.shape
        return x, ops.AcFeatureSpec(1, [(x, F, 0, F)], T=R, T_phys=R)


class R_Actor(_FusedNet):
    def __init__(self, input_shape, args):
        super().__init__(args)
        self._gain = args.gain
        self._use_orthogonal = args.use_orthogonal
        self._use_policy_active_masks = args.use_policy_active_masks
        self._use_recurrent_policy = args.use_recurrent_policy
        self._recurrent_N = args.recurrent_N
        self.tpdv = dict(dtype=torch.float32, device=self.device)
        self.base = MLPBase(args, input_shape)
        self.rnn = RNNLayer(self.rnn_hidden_dim, self.rnn_hidden_dim, self._recurrent_N, self._use_orthogonal)
        self.act = ACTLayer(args.n_actions, self.rnn_hidden_dim, self._use_orthogonal, self._gain)

    def forward(self, obs, rnn_states, available_actions=None, deterministic=False, q_noise=None):
        """obs [B,1,F], rnn_states [1,B,M], available_actions [B,1,n_act] ->
        (actions [B,1,1] int64, action_log_probs [B,1], rnn_states [B,1,M])  (ippo_actor.py:43-72)."""
        x, spec = self._spec(obs)
        B = x.shape[0]
        h = check(rnn_states).to(**self.tpdv).reshape(B, self.rnn_hidden_dim).contiguous()
        avail = None
        if available_actions is not None:
            avail = check(available_actions).to(device=self.device).reshape(B, -1).to(torch.int32).contiguous()
        if not deterministic and q_noise is None:
            q_noise = torch.empty(1, B, self._n_actions, **self.tpdv).exponential_()
        o = ops.ac_forward(self._single(), None, 0, spec, B, 1, h_actor=h, h_strides=(0, self.rnn_hidden_dim),
                           avail=avail, avail_strides=(0, self._n_actions), mode=0 if deterministic else 1,
                           q_noise=None if deterministic else q_noise.reshape(1, B, -1).contiguous(),
                           n_actions=self._n_actions)
        return o["actions"][0].reshape(B, 1, 1), o["logp"][0].reshape(B, 1), o["h_actor"][0].reshape(B, 1, -1)

    def evaluate_actions(self, obs, rnn_states, action, available_actions=None):
        """-> (action_log_probs [R,1], dist_entropy scalar = unmasked mean)  (ippo_actor.py:74-102)."""
        from ...learners.ac_function import ActorEvalFunction
        x, spec = self._spec(obs)
        R = x.shape[0]
        h = check(rnn_states).to(**self.tpdv).reshape(R, self.rnn_hidden_dim).contiguous()
        act = check(action).to(device=self.device).reshape(R).long().contiguous()
        avail = None
        if available_actions is not None:
            avail = check(available_actions).to(device=self.device).reshape(R, -1).to(torch.int32).contiguous()
        logp, ent = ActorEvalFunction.apply(self._single(), spec, h, act, avail, self._n_actions,
                                            *list(self.parameters()))
        return logp.reshape(R, 1), ent
