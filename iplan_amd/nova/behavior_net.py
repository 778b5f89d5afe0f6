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
        if hidden_size != 64:
            raise NotImplementedError("behaviour decoder kernels are built for decoder_rnn_dim = 64")
        self.output_size, self.hidden_size, self.p = output_size, hidden_size, dropout
        self.decoder = DecoderRNN(input_size, hidden_size, output_size, num_layers, dropout)
        self._enc_stub = None

    def forward(self, curr_history, prev_latent, hidden, keep=None):
        """curr_history [E,N,L,d], prev_latent [E,N,Z], hidden [1,E*N,64] -> (pred [E*N,L,d], hidden [1,E*N,64])
        (nova/behavior_net.py:55-69; inference of one window -- training goes through Behavior_policy.learn).
        Dropout is active in train() mode, as in the reference (which never switches these modules to eval)."""
        E, N, Lw, d = curr_history.shape
        dev = curr_history.device
        arena = self._single(dev)
        if self._enc_stub is None or self._enc_stub.data.device != torch.device(dev):
            # the fused kernel stages the encoder weights too; the single-window mode never uses them
            self._enc_stub = ParamArena([EncoderRNN(d, 32, prev_latent.shape[-1], 1)], dev)
        p = self.p if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (p > 0 and keep is None) else 0
        pred, hout = ops.bdec_forward(self._enc_stub, arena, curr_history.float().reshape(1, E * N, Lw, d).contiguous(),
                                      prev_latent.float().reshape(1, E * N, -1).contiguous(),
                                      hidden.float().reshape(1, E * N, self.hidden_size).contiguous(), drop_p=p, keep=keep, seed=seed)
        return pred[0], hout
