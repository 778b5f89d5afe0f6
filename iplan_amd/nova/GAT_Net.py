"""GAT_Net -- instant-incentive GAT-RNN encoder (mirror of the reference's nova/GAT_Net.py:6-142).

Same constructor, same ``forward(obs, hidden_state)`` contract, same ``state_dict`` keys
(``encoding.*``, ``hard_bi_GRU.*_l0[_reverse]``, ``hard_encoding.*``, ``q/k/v.*``, ``rnn.*``) and --
because the sub-modules are registered in the same order with torch's default initialisers -- the
same initial weights under the same seed.  The torch sub-modules are parameter containers only:
the arithmetic is the fused HIP kernel ``iplan_gat_fwd`` / ``iplan_gat_bwd``.
"""
import torch
import torch.nn as nn

from ..arena import ParamArena


def gumbel_noise(shape, device):
    """Gumbel(0, 1) samples, the noise F.gumbel_softmax draws (`-empty_like(logits).exponential_().log()`, nova/GAT_Net.py:93).
    On the GPU: one launch of iplan_gumbel_noise (counter based; seeded from torch's CPU generator, so torch.manual_seed
    makes it reproducible) -- one pass over the tensor instead of the three of the torch expression, which were 1.5 % of a
    training cycle's kernel time for the 346 MB a config-3 rollout draws.  On the CPU (host-emulated tests): the torch
    expression.  Parity tests inject the noise instead of drawing it."""
    dev = torch.device(device)
    n = 1
    for k in shape:
        n *= int(k)
    if dev.type != "cuda" or n % 4:
        return -torch.empty(shape, dtype=torch.float32, device=device).exponential_().log()
    import ctypes as C
    from .. import _lib as L
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    rc = L.get_lib().c.iplan_gumbel_noise(C.c_void_p(out.data_ptr()), C.c_int64(n), C.c_uint64(seed), C.c_void_p(L.current_stream(dev)))
    if rc != 0:
        raise L.IplanError("iplan_gumbel_noise: " + L.get_lib().c.iplan_last_error().decode())
    return out


class GAT_Net(nn.Module):
    def __init__(self, input_shape, args):
        super().__init__()
        self.args = args
        self.input_shape = input_shape
        self.max_vehicle_num = args.max_vehicle_num
        self.rnn_hidden_dim = args.GAT_hidden_dim
        self.attention_dim = args.attention_dim
        H, A = self.rnn_hidden_dim, self.attention_dim
        self.encoding = nn.Linear(input_shape, H)
        self.hard_bi_GRU = nn.GRU(H * 2, H, bidirectional=True)
        self.hard_encoding = nn.Linear(H * 2, 2)
        self.q = nn.Linear(H, A, bias=False)
        self.k = nn.Linear(H, A, bias=False)
        self.v = nn.Linear(H, A)
        self.rnn = nn.GRUCell(A, A)
        self._arena = None      # set by the owning policy (stacked over agents) or lazily (n_nets = 1)
        self._net = 0

    def attach(self, arena, net):
        self._arena, self._net = arena, net

    def _own_arena(self, device):
        if self._arena is None or self._arena.data.device != torch.device(device):
            self._arena = ParamArena([self], device)
            self._net = 0
        return self._arena

    def forward(self, obs, hidden_state, noise=None):
        """obs [B, N, D], hidden_state [B*N, A] -> [B*N, A]  (nova/GAT_Net.py:41-142).
        ``noise`` ([B*N*(N-1), 2] gumbel samples) may be injected; by default it is drawn from
        torch's generator on the input's device exactly where the reference draws it."""
        B, N, D = obs.shape
        A = self.attention_dim
        arena = self._own_arena(obs.device)
        if noise is None:
            noise = gumbel_noise((B * N * (N - 1), 2), obs.device)
        if arena.n_nets != 1:
            view = _SingleNetView(arena, self._net)
        else:
            view = arena
        from .gat_function import GatFunction
        out = GatFunction.apply(view, obs.float().contiguous(), hidden_state.float().reshape(B, N, A).contiguous(),
                                noise.float().reshape(1, B, N, N - 1, 2).contiguous(),
                                *[p for p in self.parameters()])
        return out.reshape(B * N, A)


class _SingleNetView:
    """Presents net `i` of a stacked arena as a 1-net arena (no copy)."""

    def __init__(self, arena, i):
        self.data = arena.data[i:i + 1]
        self.grad = arena.grad[i:i + 1]
        self.n_nets = 1
        self.net_stride = arena.net_stride
        self.off = arena.off
        self.offsets = arena.offsets
        self.names = arena.names
        self.shapes = arena.shapes
        self.size = arena.size
        self.trainable = arena.trainable
        self.act_tanh = getattr(arena, "act_tanh", False)
