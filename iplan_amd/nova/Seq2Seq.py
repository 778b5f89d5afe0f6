"""Seq2Seq trajectory predictor (mirror of nova/Seq2Seq.py:6-92): same classes, constructor signatures, ``forward`` contract,
reshape helpers and ``state_dict`` keys (``encoder.rnn.*_l<k>``, ``decoder.rnn.*_l<k>``, ``decoder.linear.*``), so weights
interchange with the reference.  The reference never calls it on its training path (it is the GRIP-style baseline the GAT
replaced); it is provided for API completeness and runs inference through ``iplan_seq2seq_fwd`` (one launch for the encoder
stack and all autoregressive decoder steps).  Gradients are not implemented for this module."""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from ..arena import ParamArena


class EncoderRNN(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.rnn = nn.GRU(input_size, hidden_size, num_layers, batch_first=True)


class DecoderRNN(nn.Module):
    def __init__(self, hidden_size, output_size, num_layers, dropout=0.5):
        super().__init__()
        self.hidden_size, self.output_size, self.num_layers = hidden_size, output_size, num_layers
        self.rnn = nn.GRU(output_size, hidden_size, num_layers, batch_first=True)
        self.dropout = nn.Dropout(p=dropout)
        self.linear = nn.Linear(hidden_size, output_size)
        self.tanh = nn.Tanh()


class Seq2Seq(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, pred_length, num_node, output_size=2, dropout=0.5,
                 teacher_forcing_ratio=0.5):
        super().__init__()
        if hidden_size not in (32, 64) or not 1 <= num_layers <= 4 or input_size > 64 or output_size > 16:
            raise NotImplementedError("the Seq2Seq kernel is built for hidden_size in {32, 64}, 1-4 layers, input_size <= 64, "
                                      "output_size <= 16")
        self.pred_length = pred_length
        self.teacher_forcing_ratio = teacher_forcing_ratio
        self.num_node = num_node
        self.encoder = EncoderRNN(input_size, hidden_size, num_layers)
        self.decoder = DecoderRNN(hidden_size, output_size, num_layers, dropout)
        self._arena = None

    def _own_arena(self, device):
        if self._arena is None or self._arena.data.device != torch.device(device):
            self._arena = ParamArena([self], device)
        return self._arena

    def forward(self, in_data, last_location, teacher_location=None, keep=None):
        """in_data (N*V, T, C), last_location (N*V, 1, O), teacher_location (N*V, pred_length, O) or None -> (N*V, pred_length, O)
        (nova/Seq2Seq.py:52-70).  One ``np.random.random()`` coin per step decides teacher forcing, drawn in the reference's
        order.  Dropout is active in train() mode like the reference's; ``keep`` ([pred_length, N*V, H] keep flags) may be
        injected, otherwise it is drawn from torch's generator on the input's device."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # a training loop would get a graph-less tensor back: refuse loudly, whatever the inputs' own requires_grad
            raise NotImplementedError("iplan_amd.nova.Seq2Seq is inference only (forward kernel, no backward): call it under "
                                      "torch.no_grad() or freeze its parameters")
        dev = in_data.device
        arena = self._own_arena(dev)
        rows, T_in, In = in_data.shape
        H, layers, O, P = self.encoder.hidden_size, self.encoder.num_layers, self.decoder.output_size, self.pred_length
        coins = [np.random.random() < self.teacher_forcing_ratio for _ in range(P)]
        a = L.Seq2SeqArgs()
        a.rows, a.T_in, a.In, a.H, a.layers, a.P, a.O = rows, T_in, In, H, layers, P, O
        x = in_data.detach().to(torch.float32).contiguous()
        last = last_location.detach().to(torch.float32).reshape(rows, O).contiguous()
        a.x, a.last = x.data_ptr(), last.data_ptr()
        keepalive = [x, last]
        if teacher_location is not None and any(coins):
            tl = teacher_location.detach().to(torch.float32).reshape(rows, P, O).contiguous()
            ct = torch.as_tensor(np.array(coins, dtype=np.int32), device=dev)
            a.teacher, a.coins = tl.data_ptr(), ct.data_ptr()
            keepalive += [tl, ct]
        p = self.decoder.dropout.p if self.training else 0.0
        if p > 0:
            if keep is None:
                keep = torch.empty(P, rows, H, device=dev).bernoulli_(1.0 - p)
            keep = keep.to(device=dev, dtype=torch.float32).reshape(P, rows, H).contiguous()
            a.keep, a.drop_p = keep.data_ptr(), p
            keepalive.append(keep)
        a.params = arena.data.data_ptr()
        for k in range(layers):
            for j, nm in enumerate(("weight_ih", "weight_hh", "bias_ih", "bias_hh")):
                a.enc_off[4 * k + j] = arena.off(f"encoder.rnn.{nm}_l{k}")
                a.dec_off[4 * k + j] = arena.off(f"decoder.rnn.{nm}_l{k}")
        a.lin_off[0], a.lin_off[1] = arena.off("decoder.linear.weight"), arena.off("decoder.linear.bias")
        out = torch.empty(rows, P, O, dtype=torch.float32, device=dev)
        a.out = out.data_ptr()
        L.get_lib().call("iplan_seq2seq_fwd", a, L.current_stream(dev))
        return out

    def reshape_for_rnn(self, feature):
        N, C, T, V = feature.size()
        return feature.permute(0, 3, 2, 1).contiguous().view(N * V, T, C)

    def reshape_from_rnn(self, predicted):
        NV, T, C = predicted.size()
        return predicted.view(-1, self.num_node, T, C).permute(0, 3, 2, 1).contiguous()

    def reshape_for_context(self, feature):
        NV, H = feature.size()
        return feature.view(-1, self.num_node, H)
