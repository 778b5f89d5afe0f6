"""autograd bookkeeping for the fused GAT kernels (no arithmetic here)."""
import torch

from .. import ops


class _GradSink:
    """Arena look-alike whose ``grad`` is a private buffer: autograd accumulates the returned slices into
    the parameters' ``.grad`` (views of the real gradient arena), as torch semantics require."""

    def __init__(self, arena):
        self.grad = torch.zeros_like(arena.data)
        self.off = arena.off
        self.offsets, self.shapes, self.names = arena.offsets, arena.shapes, arena.names


def grads_to_params(sink, net=0):
    outs = []
    for k in sink.names:
        o = sink.offsets[k]
        n = int(torch.Size(sink.shapes[k]).numel())
        outs.append(sink.grad[net, o:o + n].view(sink.shapes[k]))
    return outs


class GatFunction(torch.autograd.Function):
    """Single-net GAT forward/backward.  The parameter tensors are passed as (unused) inputs so
    autograd routes their gradients; the kernels read the weights from the arena they view."""

    @staticmethod
    def forward(ctx, arena, obs, h_prev, noise, *params):
        need = any(ctx.needs_input_grad)
        out, saved = ops.gat_forward(arena, obs.unsqueeze(0), None, h_prev.unsqueeze(0), noise, save=need)
        if need:
            ctx.arena = arena
            ctx.saved_acts = saved
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        sink = _GradSink(ctx.arena)
        ops.gat_backward(sink, ctx.saved_acts, gout.contiguous().unsqueeze(0))
        return (None, None, None, None, *grads_to_params(sink))
