"""autograd bookkeeping for the fused actor / critic kernels when the modules are called directly
(R_Actor.evaluate_actions / R_Critic.forward with grad enabled).  The fused learner path
(IPPOLearner.train) bypasses autograd entirely; no arithmetic happens here."""
import torch

from .. import ops
from ..nova.gat_function import _GradSink, grads_to_params


def _mask_untrainable(arena, grads):
    trainable = getattr(arena, "trainable", None)
    return [g if trainable is None or trainable.get(k, True) else None for k, g in zip(arena.names, grads)]


class ActorEvalFunction(torch.autograd.Function):
    """R_Actor.evaluate_actions for one net: (logp [R], mean entropy scalar)."""

    @staticmethod
    def forward(ctx, arena, spec, h, actions, avail, n_actions, *params):
        R = h.shape[0]
        need = any(ctx.needs_input_grad)
        o = ops.ac_forward(arena, None, 0, spec, R, 1, h_actor=h, h_strides=(0, h.shape[1]), avail=avail,
                           avail_strides=(0, n_actions), mode=2, actions_in=actions, act_strides=(0, 1),
                           n_actions=n_actions, ksplit=1, save=need, want_entropy=True, want_h=False)
        if need:
            ctx.pack = (arena, o, R)
        return o["logp"][0], o["entropy"][0].mean()

    @staticmethod
    def backward(ctx, g_logp, g_ent):
        arena, fwd, R = ctx.pack
        sink = _GradSink(arena)
        ops.ac_backward(fwd, sink, None, g_logp=g_logp.reshape(1, R).contiguous(),
                        g_entropy=(g_ent / R).reshape(1, 1).expand(1, R).contiguous())
        return (None, None, None, None, None, None, *_mask_untrainable(arena, grads_to_params(sink)))


class CriticFunction(torch.autograd.Function):
    """R_Critic.forward for one net: (values [R], h' [R, M])."""

    @staticmethod
    def forward(ctx, arena, spec, h, *params):
        R = h.shape[0]
        need = any(ctx.needs_input_grad)
        o = ops.ac_forward(None, arena, 1, spec, R, 1, h_critic=h, h_strides=(0, h.shape[1]), ksplit=1 if need else None,
                           save=need)
        if need:
            ctx.pack = (arena, o, R)
        ctx.mark_non_differentiable(o["h_critic"][0])
        return o["values"][0], o["h_critic"][0]

    @staticmethod
    def backward(ctx, g_v, _g_h):
        arena, fwd, R = ctx.pack
        sink = _GradSink(arena)
        ops.ac_backward(fwd, None, sink, g_values=g_v.reshape(1, R).contiguous())
        return (None, None, None, *_mask_untrainable(arena, grads_to_params(sink)))
