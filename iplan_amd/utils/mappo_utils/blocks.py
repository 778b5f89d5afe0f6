"""Parameter containers with the reference's module tree (utils/mappo_utils/{mlp,rnn,act,
distributions,popart}.py) so that ``state_dict`` keys, parameter order, ``requires_grad`` flags and
seed-for-seed initial values are identical to the reference's R_Actor / R_Critic.  They hold
weights only: the arithmetic of the whole actor / critic is the fused kernel ``iplan_ac_fwd``."""
import copy
import math

import torch
import torch.nn as nn

from .util import init


def _init_fn(use_orthogonal):
    return nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_


def _zero_bias(x):
    return nn.init.constant_(x, 0)


class MLPLayer(nn.Module):
    """mlp.py:6-29.  fc_h is the unused-but-registered template the reference clones fc2 from."""

    def __init__(self, input_dim, hidden_size, layer_N, use_orthogonal, use_ReLU):
        super().__init__()
        self._layer_N = layer_N
        act = nn.ReLU if use_ReLU else nn.Tanh
        gain = nn.init.calculate_gain("relu" if use_ReLU else "tanh")
        w_init = _init_fn(use_orthogonal)

        def make(i, o):
            return nn.Sequential(init(nn.Linear(i, o), w_init, _zero_bias, gain=gain), act(), nn.LayerNorm(o))
        self.fc1 = make(input_dim, hidden_size)
        self.fc_h = make(hidden_size, hidden_size)
        self.fc2 = nn.ModuleList([copy.deepcopy(self.fc_h) for _ in range(layer_N)])


class MLPBase(nn.Module):
    """mlp.py:31-55."""

    def __init__(self, args, obs_shape):
        super().__init__()
        self._use_feature_normalization = args.use_feature_normalization
        self.hidden_size = args.mlp_hidden_dim
        if self._use_feature_normalization:
            self.feature_norm = nn.LayerNorm(obs_shape)
        self.mlp = MLPLayer(obs_shape, self.hidden_size, args.layer_N, args.use_orthogonal, args.use_ReLU)


class RNNLayer(nn.Module):
    """rnn.py:7-22."""

    def __init__(self, inputs_dim, outputs_dim, recurrent_N, use_orthogonal):
        super().__init__()
        self.rnn = nn.GRU(inputs_dim, outputs_dim, num_layers=recurrent_N, batch_first=True)
        for name, param in self.rnn.named_parameters():
            if "bias" in name:
                nn.init.constant_(param, 0)
            elif "weight" in name:
                _init_fn(use_orthogonal)(param)
        self.norm = nn.LayerNorm(outputs_dim)


class Categorical(nn.Module):
    """distributions.py:55-62."""

    def __init__(self, num_inputs, num_outputs, use_orthogonal=True, gain=0.01):
        super().__init__()
        self.linear = init(nn.Linear(num_inputs, num_outputs), _init_fn(use_orthogonal), _zero_bias, gain)


class ACTLayer(nn.Module):
    """act.py:13-18."""

    def __init__(self, n_actions, inputs_dim, use_orthogonal, gain):
        super().__init__()
        self.action_out = Categorical(inputs_dim, n_actions, use_orthogonal, gain)


class PopArt(nn.Module):
    """popart.py:7-46.  Used purely as Linear(M -> 1); the four statistics are registered
    Parameters with requires_grad=False exactly as in the reference (they sit in
    critic.parameters() and in the optimiser's param list but never change)."""

    def __init__(self, input_shape, output_shape):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(output_shape, input_shape))
        self.bias = nn.Parameter(torch.empty(output_shape))
        self.stddev = nn.Parameter(torch.ones(output_shape), requires_grad=False)
        self.mean = nn.Parameter(torch.zeros(output_shape), requires_grad=False)
        self.mean_sq = nn.Parameter(torch.zeros(output_shape), requires_grad=False)
        self.debiasing_term = nn.Parameter(torch.tensor(0.0), requires_grad=False)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(input_shape)
        nn.init.uniform_(self.bias, -bound, bound)
