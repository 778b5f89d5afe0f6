"""CPU (host-emulated kernels): the stacked three-layer perceptron of the FC ablation (csrc/mlp3.hip) against torch
autograd -- odd input / output widths, ragged row counts, both output modes and both gradient sources."""
import pytest
import torch
import torch.nn as nn

from iplan_amd import _lib as L
from iplan_amd import ops
from iplan_amd.arena import ParamArena
from tests.emu.emu_lib import get_emu_lib


@pytest.fixture(autouse=True)
def emu():
    L.use_library_for_tests(get_emu_lib())
    yield
    L.use_library_for_tests(None)


class Net(nn.Module):
    def __init__(self, k0, h, o):
        super().__init__()
        self.linear_1, self.linear_2, self.out = nn.Linear(k0, h), nn.Linear(h, h), nn.Linear(h, o)

    def forward(self, x, softmax):
        y = self.out(torch.tanh(self.linear_2(torch.tanh(self.linear_1(x)))))
        return torch.softmax(y, -1) if softmax else y


@pytest.mark.parametrize("K0,H,O,rows,softmax", [(20, 32, 8, 37, True), (58, 64, 50, 70, False), (64, 64, 64, 16, False), (5, 32, 3, 129, True)])
def test_mlp3_forward_backward_vs_autograd(K0, H, O, rows, softmax):
    torch.manual_seed(K0 + O)
    n_nets = 2
    nets = [Net(K0, H, O) for _ in range(n_nets)]
    refs = [Net(K0, H, O) for _ in range(n_nets)]
    for a, b in zip(nets, refs):
        b.load_state_dict(a.state_dict())
    arena = ParamArena(nets, "cpu")
    x = torch.randn(n_nets, rows, K0)
    g_out = torch.randn(n_nets, rows, O)
    target = torch.randn(n_nets, rows, O)
    # explicit upstream gradient
    fwd = ops.mlp3_forward(arena, "", x, H, O, softmax=softmax)
    dx = ops.mlp3_backward(arena, "", fwd, g_out=g_out, want_dx=True)
    for n in range(n_nets):
        xr = x[n].clone().requires_grad_(True)
        y = refs[n](xr, softmax)
        assert torch.allclose(fwd["out"][n], y, rtol=1e-5, atol=1e-6)
        (y * g_out[n]).sum().backward()
        assert torch.allclose(dx[n], xr.grad, rtol=1e-4, atol=1e-5)
        for name, p in refs[n].named_parameters():
            got = arena.grad_of(n, name)
            assert torch.allclose(got, p.grad, rtol=1e-4, atol=1e-4 * max(1.0, p.grad.abs().max().item())), name
            p.grad = None
    if softmax:
        return
    # L1 loss against a target: numerator from the forward, -sign(target - out) * scale from the backward, accumulated (beta = 1)
    before = arena.grad.clone()
    fwd = ops.mlp3_forward(arena, "", x, H, O, target=target)
    ops.mlp3_backward(arena, "", fwd, g_scale=0.25, beta=1.0)
    for n in range(n_nets):
        y = refs[n](x[n], False)
        assert torch.allclose(fwd["l1"][n], (target[n] - y).abs().sum(), rtol=1e-5)
        ((target[n] - y).abs().sum() * 0.25).backward()
        for name, p in refs[n].named_parameters():
            k = arena.off(name)
            got = arena.grad[n, k:k + p.numel()].view(p.shape) - before[n, k:k + p.numel()].view(p.shape)
            assert torch.allclose(got, p.grad, rtol=1e-4, atol=1e-4 * max(1.0, p.grad.abs().max().item())), name
