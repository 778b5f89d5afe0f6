"""Parameter containers of the FC behaviour ablation -- mirror of nova/behavior_FC_net.py:6-60 (same constructor
arguments, ``state_dict`` keys and default initialisation); the arithmetic runs in the stacked ``iplan_mlp3`` kernels."""
import torch
import torch.nn as nn

from .. import ops
from ..arena import ParamArena


class _Mlp3(nn.Module):
    def __init__(self, input_size, hidden_size, output_size):
        super().__init__()
        self.input_size, self.hidden_size, self.output_size = input_size, hidden_size, output_size
        self.linear_1 = nn.Linear(input_size, hidden_size)
        self.linear_2 = nn.Linear(hidden_size, hidden_size)
        self.out = nn.Linear(hidden_size, output_size)

    def _run(self, x, softmax):
        dev = x.device
        arena = ParamArena([self], dev)
        lead = x.shape[:-1]
        with torch.no_grad():
            y = ops.mlp3_forward(arena, "", x.reshape(1, -1, x.shape[-1]).to(torch.float32).contiguous(), self.hidden_size,
                                 self.output_size, softmax=softmax, save=False)["out"]
        return y.reshape(*lead, self.output_size)


class Encoder_3FC(_Mlp3):
    """nova/behavior_FC_net.py:6-20: softmax(out(tanh(linear_2(tanh(linear_1(x))))))."""

    def forward(self, input):
        return self._run(input, True)


class Decoder_3FC(_Mlp3):
    """nova/behavior_FC_net.py:23-37."""

    def forward(self, encoded_input):
        return self._run(encoded_input, False)


class LILI_Latent_Decoder(nn.Module):
    """nova/behavior_FC_net.py:40-60: Decoder_3FC on [flattened history window || latent]."""

    def __init__(self, input_size, hidden_size, output_size):
        super().__init__()
        self.decoder = Decoder_3FC(input_size, hidden_size, output_size)

    def forward(self, curr_history, prev_latent):
        n_thread, N = curr_history.shape[:2]
        x = torch.cat([curr_history.reshape(n_thread, N, -1), prev_latent.reshape(n_thread, N, -1)], dim=-1)
        return self.decoder(x)
