"""CPU (host-emulated kernel build): fused GAT backward vs the reference's own gradients
(tests/golden/gat_*.pt: fp32 grads of the real reference and fp64 grads of the oracle)."""
import pytest
import torch

from iplan_amd import _lib as L
from iplan_amd import ops
from tests.emu.emu_lib import get_emu_lib


@pytest.fixture(autouse=True)
def emu():
    L.use_library_for_tests(get_emu_lib())
    yield
    L.use_library_for_tests(None)


def check_gat_backward(g, device, tol=1e-5):
    from iplan_amd.arena import ParamArena
    from iplan_amd.config import default_args
    from iplan_amd.nova.GAT_Net import GAT_Net
    B, N, D = g["B"], g["N"], g["D"]
    args = default_args("highway", use_cuda=(device != "cpu"), max_vehicle_num=N)
    nets = [GAT_Net(D, args) for _ in range(2)]           # two stacked nets: the second is a decoy
    nets[0].load_state_dict(g["params"])
    arena = ParamArena(nets, device)
    d0 = 5
    obs = g["obs"].to(device)
    src0 = obs[..., :d0].contiguous().unsqueeze(0).expand(2, -1, -1, -1).contiguous()
    src1 = obs[..., d0:].contiguous().unsqueeze(0).expand(2, -1, -1, -1).contiguous()
    h_prev = g["h_prev"].to(device).reshape(1, B, N, 32).expand(2, -1, -1, -1).contiguous()
    noise = g["noise"].to(device).reshape(1, B, N, N - 1, 2).expand(2, -1, -1, -1, -1).contiguous()
    out, saved = ops.gat_forward(arena, src0, src1, h_prev, noise, save=True)
    assert (out[0].reshape(B * N, 32).cpu() - g["out"]).abs().max() < 1e-5
    gout = g["gout"].to(device).reshape(1, B, N, 32).expand(2, -1, -1, -1).contiguous()
    ops.gat_backward(arena, saved, gout)
    for k, ref in g["grads64"].items():
        got = arena.grad_of(0, k).cpu().double()
        err = (got - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (k, err, ref.abs().max().item())


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_gat_backward_emulated(golden, tag):
    check_gat_backward(golden("gat_" + tag), "cpu")
