"""R_Critic -- recurrent value network (mirror of modules/critics/ippo_critic.py:10-65)."""
import torch
import torch.nn as nn

from ...utils.mappo_utils.blocks import MLPBase, PopArt, RNNLayer
from ...utils.mappo_utils.util import check, init
from ..agents.ippo_actor import _FusedNet


class R_Critic(_FusedNet):
    def __init__(self, input_shape, args):
        super().__init__(args)
        self._use_orthogonal = args.use_orthogonal
        self._use_recurrent_policy = args.use_recurrent_policy
        self._recurrent_N = args.recurrent_N
        self._use_popart = args.use_popart
        self.tpdv = dict(dtype=torch.float32, device=self.device)
        w_init = nn.init.orthogonal_ if self._use_orthogonal else nn.init.xavier_uniform_
        self.base = MLPBase(args, input_shape)
        self.rnn = RNNLayer(self.rnn_hidden_dim, self.rnn_hidden_dim, self._recurrent_N, self._use_orthogonal)
        head = PopArt(self.rnn_hidden_dim, 1) if self._use_popart else nn.Linear(self.rnn_hidden_dim, 1)
        self.v_out = init(head, w_init, lambda x: nn.init.constant_(x, 0))

    def forward(self, obs, rnn_states):
        """obs [B,1,F], rnn_states [1,B,M] -> (values [B,1,1], rnn_states [B,1,M])  (ippo_critic.py:47-65)."""
        from ...learners.ac_function import CriticFunction
        x, spec = self._spec(obs)
        B = x.shape[0]
        h = check(rnn_states).to(**self.tpdv).reshape(B, self.rnn_hidden_dim).contiguous()
        v, hn = CriticFunction.apply(self._single(), spec, h, *list(self.parameters()))
        return v.reshape(B, 1, 1), hn.reshape(B, 1, -1)
