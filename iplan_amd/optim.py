"""FusedAdam: torch.optim.Adam semantics (weight_decay 0, amsgrad False) on arena slices, with the
preceding ``clip_grad_norm_`` folded into the same launch sequence and all nets of an arena
updatable in one launch.

One FusedAdam object per agent keeps the reference's API (``learner.actor_optimizers[i]``,
``.zero_grad()``, ``.step()``, ``.state_dict()`` / ``.load_state_dict()`` in torch.optim.Adam's
checkpoint format, ``param_groups[0]['lr']`` for the linear LR decay); objects created over the
same arenas share moment buffers, and ``step_all`` advances every agent in one launch.
"""
import ctypes as C
import math

import torch

from . import _lib as L

def _moments(arena):
    """(exp_avg, exp_avg_sq) twins of an arena, shared by the per-agent optimisers over it; they live ON the arena object,
    so they are released with it."""
    mv = getattr(arena, "_adam_moments", None)
    if mv is None:
        mv = arena._adam_moments = (torch.zeros_like(arena.data), torch.zeros_like(arena.data))
    return mv


def grad_sqnorm(arena, nets, out, slot, lib=None):
    """out[k, slot] = ||grad of net nets[0]+k||^2 for the contiguous net range `nets`."""
    lib = lib or L.get_lib()
    n0, cnt = nets
    rc = lib.c.iplan_grad_sqnorm(C.c_void_p(arena.grad.data_ptr() + 4 * n0 * arena.net_stride),
                                 C.c_int64(arena.net_stride), C.c_int64(0), C.c_int64(arena.size), C.c_int32(cnt),
                                 C.c_void_p(out.data_ptr()), C.c_int32(out.stride(0)), C.c_int32(slot),
                                 C.c_void_p(L.current_stream(arena.data.device)))
    if rc != 0:
        raise L.IplanError("iplan_grad_sqnorm: " + lib.c.iplan_last_error().decode())


def adam_launch(arena, nets, steps, lr, betas, eps, sqnorm, slot, max_norm, write_clipped=True, weight_decay=0.0, lib=None):
    lib = lib or L.get_lib()
    m, v = _moments(arena)
    n0, cnt = nets
    a = L.AdamArgs()
    shift = 4 * n0 * arena.net_stride
    a.param = arena.data.data_ptr() + shift
    a.grad = arena.grad.data_ptr() + shift
    a.exp_avg = m.data_ptr() + shift
    a.exp_avg_sq = v.data_ptr() + shift
    a.stride, a.off, a.n, a.n_nets = arena.net_stride, 0, arena.size, cnt
    if sqnorm is not None:
        a.sqnorm = sqnorm.data_ptr()
        a.sqnorm_stride, a.sqnorm_slot = sqnorm.stride(0), slot
    a.max_norm = max_norm
    a.write_clipped = 1 if write_clipped else 0
    a.lr, a.beta1, a.beta2, a.eps = lr, betas[0], betas[1], eps
    a.weight_decay = weight_decay
    for k in range(cnt):
        a.bc1[k] = 1.0 - betas[0] ** steps[k]
        a.bc2_sqrt[k] = math.sqrt(1.0 - betas[1] ** steps[k])
    lib.call("iplan_adam_step", a, L.current_stream(arena.data.device))
    arena.version += 1                                      # derived caches (ops.Fc1Pack) repack on the next use


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, slices, lr, eps=1e-8, weight_decay=0, betas=(0.9, 0.999)):
        """slices: list of (arena, net) whose parameters this optimiser owns, in param order."""
        self.slices = slices
        params = []
        for arena, net in slices:
            params += list(arena.modules[net].parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._steps = 0
        dev = slices[0][0].data.device
        self._sq = torch.zeros(1, len(slices), dtype=torch.float32, device=dev)
        self.last_sqnorms = self._sq

    def zero_grad(self, set_to_none=False):
        for arena, net in self.slices:
            arena.grad[net].zero_()

    @torch.no_grad()
    def step(self, max_norm=None, closure=None):
        """clip each slice's gradient to ``max_norm`` (None = no clipping) and apply Adam."""
        self._steps += 1
        g = self.param_groups[0]
        for k, (arena, net) in enumerate(self.slices):
            sq = None
            if max_norm is not None:
                grad_sqnorm(arena, (net, 1), self._sq, k)
                sq = self._sq
            adam_launch(arena, (net, 1), [self._steps], g["lr"], g["betas"], g["eps"], sq, k,
                        max_norm if max_norm is not None else 0.0, weight_decay=g["weight_decay"])

    def grad_norms(self):
        """Pre-clip L2 norms of the slices (device tensor; read it lazily to avoid a sync)."""
        return self._sq[0].sqrt()

    # -- torch.optim.Adam checkpoint format ---------------------------------------------------
    def _param_moments(self):
        out = []
        for arena, net in self.slices:
            m, v = _moments(arena)
            for name in arena.names:
                o = arena.offsets[name]
                n = int(torch.Size(arena.shapes[name]).numel())
                out.append((arena.trainable[name], m[net, o:o + n].view(arena.shapes[name]),
                            v[net, o:o + n].view(arena.shapes[name])))
        return out

    def state_dict(self):
        state = {}
        if self._steps > 0:
            pm = self._param_moments()
            # torch.optim.Adam only creates state for parameters that ever received a gradient (the unused `fc_h`
            # template layer of MLPLayer never does).  Every arena tensor has a gradient view here, so "received a
            # gradient" is read off the second moment (non-zero somewhere); ONE host read-back for all tensors.  A
            # parameter whose gradients were exactly zero at every step is written without state, which torch loads as
            # "state not created yet" -- the same values it would have held (all-zero moments).
            live = torch.stack([(v != 0).any() if trainable else torch.zeros((), dtype=torch.bool, device=v.device)
                                for trainable, _, v in pm]).cpu().tolist()
            for idx, (trainable, m, v) in enumerate(pm):
                if live[idx]:
                    state[idx] = {"step": torch.tensor(float(self._steps)), "exp_avg": m.detach().clone(),
                                  "exp_avg_sq": v.detach().clone()}
        g = dict(self.param_groups[0])
        g["params"] = list(range(len(g["params"])))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        pm = self._param_moments()
        for _, m, v in pm:                          # parameters absent from the file have no state: zero moments
            m.zero_()
            v.zero_()
        self._steps = 0
        for idx, st in sd["state"].items():
            _, m, v = pm[int(idx)]
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            self._steps = int(float(st["step"]))
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in sd["param_groups"][0]:
                self.param_groups[0][k] = sd["param_groups"][0][k]


def step_all(optimizers, max_norm, slices=None, steps=None, sq=None):
    """Advance the per-agent optimisers of a learner together: one norm launch and one Adam launch
    per arena for ALL agents (they must own net i of the same arenas, i = position in the list).
    ``slices`` restricts the call to some of the optimisers' arenas (indices into ``FusedAdam.slices``); a later call for
    the remaining arenas of the SAME optimiser step passes the ``steps`` this one returned through ``last_steps``.
    ``sq``: the [n_agents, n_slices] buffer the squared norms go to (default: one buffer owned by the first optimiser -- a
    caller that runs two halves of a step on different streams gives the second half its own)."""
    first = optimizers[0]
    n = len(optimizers)
    if steps is None:
        for o in optimizers:
            o._steps += 1
        steps = [o._steps for o in optimizers]
    first.last_steps = steps
    g = first.param_groups[0]
    dev = first.slices[0][0].data.device
    if sq is None:
        if getattr(first, "_sq_all", None) is None or first._sq_all.shape[0] != n:
            first._sq_all = torch.zeros(n, len(first.slices), dtype=torch.float32, device=dev)
        sq = first._sq_all
    assert sq.shape == (n, len(first.slices)) and sq.dtype == torch.float32 and sq.is_contiguous()
    for k, (arena, _) in enumerate(first.slices):
        if slices is not None and k not in slices:
            continue
        if max_norm is not None:
            grad_sqnorm(arena, (0, n), sq, k)
        adam_launch(arena, (0, n), steps, g["lr"], g["betas"], g["eps"], sq if max_norm is not None else None, k,
                    max_norm if max_norm is not None else 0.0, weight_decay=g["weight_decay"])
    return sq      # [n_agents, n_slices] squared pre-clip norms
