"""Behaviour-intent encoder / decoder networks (mirror of nova/behavior_net.py:6-69).

Parameter containers with the reference's ``state_dict`` keys; arithmetic in
``iplan_enc_fwd`` (rollout) and the fused behaviour-learning kernels (training)."""
import torch
import torch.nn as nn

from .. import ops
from ..arena import ParamArena


class _ArenaModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._arena = None
        self._net = 0

    def attach(self, arena, net):
        self._arena, self._net = arena, net

    def _single(self, device):
        from .GAT_Net import _SingleNetView
        if self._arena is None or self._arena.data.device != torch.device(device):
            self._arena = ParamArena([self], device)
            self._net = 0
        return self._arena if self._arena.n_nets == 1 else _SingleNetView(self._arena, self._net)


class EncoderRNN(_ArenaModule):
    def __init__(self, input_size, hidden_size, output_size, num_layers):
        super().__init__()
        if num_layers != 1 or hidden_size != 32:
            raise NotImplementedError("encoder kernel is built for num_encoder_layer=1, encoder_rnn_dim=32")
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.output_size = output_size
        self.linear = nn.Linear(input_size, hidden_size)
        self.rnn = nn.GRU(hidden_size, hidden_size, num_layers, batch_first=True)
        self.out = nn.Linear(hidden_size, output_size)

    def forward(self, input, hidden):
        """input [R, L, d], hidden [1, R, Rdim] -> (None, new_hidden [1, R, Rdim], latent [R, Z] softmax)
        (nova/behavior_net.py:17-22).  The per-step GRU outputs (first return value of the
        reference) are consumed by no caller on the path and are not materialised (inference)."""
        R, Lw, d = input.shape
        arena = self._single(input.device)
        x = input.float().reshape(1, R, 1, Lw, d)
        h0 = hidden.float().reshape(1, R, 1, self.hidden_size)
        lat, hL = ops.enc_forward(arena, x, h0, None, 0.0, self.output_size)
        return None, hL.reshape(1, R, self.hidden_size), lat.reshape(R, self.output_size)


class DecoderRNN(nn.Module):
    def __init__(self, input_size, hidden_size, output_size, num_layers, dropout=0.5):
        super().__init__()
        if num_layers != 1:
            raise NotImplementedError("decoder kernels are built for one GRU layer")
        self.hidden_size, self.output_size, self.num_layers = hidden_size, output_size, num_layers
        self.linear = nn.Linear(input_size, hidden_size)
        self.rnn = nn.GRU(hidden_size, hidden_size, num_layers, batch_first=True)
        self.dropout = nn.Dropout(p=dropout)
        self.out = nn.Linear(hidden_size, output_size)
        self.tanh = nn.Tanh()


class Behavior_Latent_Decoder(_ArenaModule):
    def __init__(self, input_size, hidden_size, num_layers, output_size, dropout=0.5):
        super().__init__()
        self.decoder = DecoderRNN(input_size, hidden_size, output_size, num_layers, dropout)
