"""autograd bookkeeping for the fused GAT kernels (no arithmetic here)."""
import torch

from .. import ops


class GatFunction(torch.autograd.Function):
    """Single-net GAT forward/backward.  The parameter tensors are passed as (unused) inputs so
    autograd routes their gradients; the kernels read the weights from the arena they view."""

    @staticmethod
    def forward(ctx, arena, obs, h_prev, noise, *params):
        need = any(p.requires_grad for p in params) and torch.is_grad_enabled()
        d = obs.shape[-1]
        out, saved = ops.gat_forward(arena, obs.unsqueeze(0), None, h_prev.unsqueeze(0), noise, save=need)
        if need:
            ctx.arena = arena
            ctx.saved_acts = saved
            ctx.save_for_backward(obs, h_prev)
        ctx.nparams = len(params)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        obs, h_prev = ctx.saved_tensors
        grads = ops.gat_backward(ctx.arena, obs.unsqueeze(0), None, h_prev.unsqueeze(0), ctx.saved_acts,
                                 gout.contiguous().unsqueeze(0))
        # grads: flat [1, P] gradient in arena layout -> per-parameter views
        arena = ctx.arena
        outs = []
        for k in arena.names:
            o = arena.offsets[k]
            n = int(torch.Size(arena.shapes[k]).numel())
            outs.append(grads[0, o:o + n].view(arena.shapes[k]))
        return (None, None, None, None, *outs)
