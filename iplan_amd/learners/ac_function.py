"""autograd bookkeeping for the fused actor / critic kernels (no arithmetic here)."""
import torch

from .. import ops


def _grads_to_params(arena, flat):
    outs = []
    for k in arena.names:
        o = arena.offsets[k]
        n = int(torch.Size(arena.shapes[k]).numel())
        outs.append(flat[0, o:o + n].view(arena.shapes[k]) if arena.trainable_mask.get(k, True) else None)
    return outs


class ActorEvalFunction(torch.autograd.Function):
    """R_Actor.evaluate_actions for one net: (logp [R], mean entropy scalar)."""

    @staticmethod
    def forward(ctx, arena, spec, h, actions, avail, n_actions, *params):
        R = h.shape[0]
        need = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        o = ops.ac_forward(arena, None, 0, spec, R, 1, h_actor=h, h_strides=(0, h.shape[1]), avail=avail,
                           avail_strides=(0, n_actions), mode=2, actions_in=actions, act_strides=(0, 1),
                           n_actions=n_actions, save=need, want_probs=need, want_entropy=True, want_h=False)
        if need:
            ctx.pack = (arena, spec, h, actions, avail, n_actions, o["saved"], o["probs"])
        return o["logp"][0], o["entropy"][0].mean()

    @staticmethod
    def backward(ctx, g_logp, g_ent):
        arena, spec, h, actions, avail, n_actions, saved, probs = ctx.pack
        R = h.shape[0]
        g_lp = g_logp.reshape(1, R).contiguous()
        g_e = (g_ent / R).expand(1, R).contiguous()
        flat = ops.ac_backward(arena, None, 0, spec, R, 1, saved, h_actor=h, h_strides=(0, h.shape[1]),
                               probs=probs, actions=actions.reshape(1, R), g_logp=g_lp, g_entropy=g_e,
                               n_actions=n_actions)["actor_grad"]
        return (None, None, None, None, None, None, *_grads_to_params(arena, flat))


class CriticFunction(torch.autograd.Function):
    """R_Critic.forward for one net: (values [R], h' [R, M])."""

    @staticmethod
    def forward(ctx, arena, spec, h, *params):
        R = h.shape[0]
        need = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        o = ops.ac_forward(None, arena, 1, spec, R, 1, h_critic=h, h_strides=(0, h.shape[1]), save=need)
        if need:
            ctx.pack = (arena, spec, h, o["saved"])
        ctx.mark_non_differentiable(o["h_critic"])
        return o["values"][0], o["h_critic"][0]

    @staticmethod
    def backward(ctx, g_v, _g_h):
        arena, spec, h, saved = ctx.pack
        R = h.shape[0]
        flat = ops.ac_backward(None, arena, 1, spec, R, 1, saved, h_critic=h, h_strides=(0, h.shape[1]),
                               g_values=g_v.reshape(1, R).contiguous())["critic_grad"]
        return (None, None, None, *_grads_to_params(arena, flat))
