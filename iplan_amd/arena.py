"""Stacked parameter arenas.

The reference keeps ``n_agents`` private copies of every network as separate nn.Modules and loops
over them in Python.  Here the modules still exist (same classes, same ``state_dict`` keys, so
checkpoints interchange), but their Parameters are *views* into one contiguous fp32 arena
``[n_nets, P]`` (and their ``.grad`` into a twin gradient arena).  Consequences:
  * one kernel launch reads every net's weights via (arena pointer, net stride, per-tensor offset);
  * clip_grad_norm_ / Adam / zero_grad are single flat-buffer kernels instead of per-tensor loops;
  * the data-parallel gradient all-reduce is ONE RCCL call on the gradient arena.
"""
import torch


class ParamArena:
    def __init__(self, modules, device):
        """modules: list of structurally identical nn.Modules (one per net)."""
        self.modules = list(modules)
        self.n_nets = len(self.modules)
        named = list(self.modules[0].named_parameters())
        self.names = [k for k, _ in named]
        self.shapes = {k: tuple(v.shape) for k, v in named}
        self.offsets = {}
        off = 0
        for k, v in named:
            self.offsets[k] = off
            off += (v.numel() + 3) // 4 * 4           # keep every tensor 16-byte aligned
        self.size = off
        self.trainable = {k: v.requires_grad for k, v in named}
        # bumped by every KERNEL-side write to ``data`` (Adam, collectives), by load_state_dict and by touch().  Torch-side
        # in-place writes through a Parameter bump THAT Parameter's ``_version`` (``p.data = view`` below gives it its own
        # counter, not the arena tensor's): derived caches (ops.Fc1Pack) watch both
        self.version = 0
        # MLPBase activation of an actor / critic stack (mlp.py:10: tanh when args.use_ReLU is off): read by ops.ac_forward
        self.act_tanh = bool(getattr(self.modules[0], "act_tanh", False))
        self.data = torch.zeros(self.n_nets, self.size, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.data)
        for i, m in enumerate(self.modules):
            for k, p in m.named_parameters():
                view = self.data[i, self.offsets[k]:self.offsets[k] + p.numel()].view(p.shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = self.grad[i, self.offsets[k]:self.offsets[k] + p.numel()].view(p.shape) \
                    if p.requires_grad else None
            # module.load_state_dict copies into the views through ``.data`` (its own version counter): tell derived caches
            m.register_load_state_dict_post_hook(lambda module, incompatible_keys: self._bump())

    def _bump(self):
        self.version += 1

    @staticmethod
    def colocate_grads(arenas):
        """Re-home the gradient arenas of ``arenas`` into ONE contiguous fp32 buffer (in the given order) and rebind every
        Parameter's ``.grad`` view: arenas whose gradients are always exchanged together (actor + critic of a PPO step; GAT +
        prediction decoder) then cost one collective per optimiser step instead of one each -- the exchange is latency-bound
        at these sizes (0.3 - 4 MB), so the count is what matters (DataParallel.all_reduce_grads).  Call it right after
        construction, before anything holds a pointer into ``.grad``.  Returns the flat buffer."""
        dev = arenas[0].grad.device
        flat = torch.zeros(sum(a.grad.numel() for a in arenas), dtype=torch.float32, device=dev)
        o = 0
        for a in arenas:
            n = a.grad.numel()
            g = flat[o:o + n].view(a.n_nets, a.size)
            g.copy_(a.grad)
            a.grad = g
            for i, m in enumerate(a.modules):
                for k, p in m.named_parameters():
                    if p.requires_grad:
                        p.grad = g[i, a.offsets[k]:a.offsets[k] + p.numel()].view(p.shape)
            o += n
        group = (flat, tuple(arenas))
        for a in arenas:
            a._grad_group = group
        return flat

    def touch(self):
        """Tell derived caches that ``data`` was written through a path they cannot see (``p.data.mul_()``, a raw pointer)."""
        self.version += 1

    @property
    def net_stride(self):
        return self.size

    def off(self, name):
        return self.offsets[name]

    def ranges(self, names=None):
        """(offset, numel) of each named tensor (default: all), in arena order."""
        names = self.names if names is None else names
        return [(self.offsets[k], int(torch.Size(self.shapes[k]).numel())) for k in names]

    def param(self, net, name):
        n = int(torch.Size(self.shapes[name]).numel())
        return self.data[net, self.offsets[name]:self.offsets[name] + n].view(self.shapes[name])

    def grad_of(self, net, name):
        n = int(torch.Size(self.shapes[name]).numel())
        return self.grad[net, self.offsets[name]:self.offsets[name] + n].view(self.shapes[name])
