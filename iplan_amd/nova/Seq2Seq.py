"""Seq2Seq trajectory predictor (mirror of nova/Seq2Seq.py:6-92): same classes, constructor signatures, ``forward`` contract,
reshape helpers and ``state_dict`` keys (``encoder.rnn.*_l<k>``, ``decoder.rnn.*_l<k>``, ``decoder.linear.*``), so weights
interchange with the reference.  The reference never calls it on its training path (it is the GRIP-style baseline the GAT
replaced); it is provided for API completeness: ``iplan_seq2seq_fwd`` runs the encoder stack and all autoregressive decoder
steps in one launch, and when gradients are wanted the forward saves its gate record and ``iplan_seq2seq_bwd`` + ``iplan_wgrad``
produce the parameter gradients (BPTT through the prediction feedback, like autograd through the reference's loop).  The data
inputs (in_data, last_location, teacher_location) get no gradient."""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..arena import ParamArena
from .gat_function import _GradSink, grads_to_params


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
        dev = in_data.device
        # Gradients flow to the PARAMETERS only (iplan_seq2seq_bwd + iplan_wgrad).  In the reference (plain nn.GRU / nn.Linear
        # autograd, nova/Seq2Seq.py:41-70) they would also reach in_data / last_location; a caller that feeds this module from a
        # trainable upstream module must hear about the difference instead of silently training the upstream on zeros.
        if torch.is_grad_enabled():
            for nm, t in (("in_data", in_data), ("last_location", last_location), ("teacher_location", teacher_location)):
                if isinstance(t, torch.Tensor) and t.requires_grad:
                    raise NotImplementedError(f"Seq2Seq.forward: {nm}.requires_grad is set, but this implementation differentiates with respect "
                                              "to the module's parameters only (no input gradients); detach the input or run under no_grad")
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
        params = [dict(self.named_parameters())[k] for k in arena.names]
        if torch.is_grad_enabled() and any(q.requires_grad for q in params):
            return _Seq2SeqFunction.apply(arena, a, keepalive, *params)
        return _launch_fwd(a, dev, None)

    def reshape_for_rnn(self, feature):
        N, C, T, V = feature.size()
        return feature.permute(0, 3, 2, 1).contiguous().view(N * V, T, C)

    def reshape_from_rnn(self, predicted):
        NV, T, C = predicted.size()
        return predicted.view(-1, self.num_node, T, C).permute(0, 3, 2, 1).contiguous()

    def reshape_for_context(self, feature):
        NV, H = feature.size()
        return feature.view(-1, self.num_node, H)


def save_floats(rows, T_in, P, layers, H):
    """floats of IplanSeq2SeqArgs.save / IplanSeq2SeqBwdArgs.dsave (include/iplan_hip.h)"""
    steps = (T_in + P) * layers * rows
    return steps * 6 * H + P * rows * (16 + H), steps * 4 * H + P * rows * 16


def _launch_fwd(a, dev, save):
    out = torch.empty(a.rows, a.P, a.O, dtype=torch.float32, device=dev)
    a.out = out.data_ptr()
    a.save = save.data_ptr() if save is not None else None
    L.get_lib().call("iplan_seq2seq_fwd", a, L.current_stream(dev))
    return out


def seq2seq_backward(sink, a, save, g_out):
    """Parameter gradients of a saving forward launch ``a`` into ``sink.grad`` [1, arena floats]: the BPTT launch writes the
    row-level gate gradients, eight-ish contractions per layer fold them into the weights (the autograd of nn.GRU / nn.Linear
    under a loss on nova/Seq2Seq.py:70's output)."""
    dev = save.device
    rows, T_in, In, H, layers, P, O = a.rows, a.T_in, a.In, a.H, a.layers, a.P, a.O
    n_save, n_dsave = save_floats(rows, T_in, P, layers, H)
    dsave = torch.empty(n_dsave, dtype=torch.float32, device=dev)
    b = L.Seq2SeqBwdArgs()
    b.fwd = a
    g = g_out.to(torch.float32).contiguous()
    b.g_out, b.dsave = g.data_ptr(), dsave.data_ptr()
    L.get_lib().call("iplan_seq2seq_bwd", b, L.current_stream(dev))
    w = ops.Wgrad(sink.grad, 1, tag="iplan_wgrad_seq2seq")
    off = sink.off
    sp, dp = save.data_ptr(), dsave.data_ptr()
    S6, S4 = 6 * H, 4 * H
    step6, step4 = layers * rows * S6, layers * rows * S4
    dec_rec = sp + 4 * (T_in + P) * step6                      # [P, rows, 16 + H]
    ddec = dp + 4 * (T_in + P) * step4                         # [P, rows, 16]
    for stack, tau0, n_t in (("encoder", 0, T_in), ("decoder", T_in, P)):
        for k in range(layers):
            dy = dp + 4 * ((tau0 * layers + k) * rows * S4)
            rec = sp + 4 * ((tau0 * layers + k) * rows * S6)
            if k:                                              # the layer below's h_new of the same step
                xin, xs, K = sp + 4 * ((tau0 * layers + k - 1) * rows * S6 + 5 * H), (0, step6, S6), H
            elif stack == "encoder":
                xin, xs, K = a.x, (0, In, T_in * In), In       # in_data [rows, T_in, In] read as (step, row)
            else:
                xin, xs, K = dec_rec, (0, rows * (16 + H), 16 + H), O
            w.add(dy, (0, step4, S4), 3 * H, n_t, rows, x=xin, x_strides=xs, K=K,
                  dw_off=off(f"{stack}.rnn.weight_ih_l{k}"), db_off=off(f"{stack}.rnn.bias_ih_l{k}"))
            w.add(dy, (0, step4, S4), 3 * H, n_t, rows, x=rec, x_strides=(0, step6, S6), K=H,
                  dw_off=off(f"{stack}.rnn.weight_hh_l{k}"), db_off=off(f"{stack}.rnn.bias_hh_l{k}"), seg=(2 * H, 0, 3 * H))
    w.add(ddec, (0, rows * 16, 16), O, P, rows, x=dec_rec + 4 * 16, x_strides=(0, rows * (16 + H), 16 + H), K=H,
          dw_off=off("decoder.linear.weight"), db_off=off("decoder.linear.bias"))
    w._keep += [save, dsave, g]
    w.run()


class _Seq2SeqFunction(torch.autograd.Function):
    """The parameter tensors are passed as (unused) inputs so autograd routes their gradients; the kernels read the weights
    from the arena they view (same bookkeeping as nova/gat_function.py)."""

    @staticmethod
    def forward(ctx, arena, a, keepalive, *params):
        dev = arena.data.device
        n_save, _ = save_floats(a.rows, a.T_in, a.P, a.layers, a.H)
        save = torch.empty(n_save, dtype=torch.float32, device=dev)
        out = _launch_fwd(a, dev, save)
        ctx.arena, ctx.a, ctx.keepalive, ctx.rec = arena, a, keepalive, save
        return out

    @staticmethod
    def backward(ctx, g_out):
        sink = _GradSink(ctx.arena)
        seq2seq_backward(sink, ctx.a, ctx.rec, g_out)
        return (None, None, None, *grads_to_params(sink))
