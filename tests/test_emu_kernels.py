"""CPU: the UNMODIFIED kernel sources (iplan_amd/csrc/*.hip), compiled against the host emulator
shim (tests/emu), reproduce the committed reference outputs.  This exercises the real index math,
MFMA fragment layouts, LDS hand-offs and barrier structure of the HIP kernels without a GPU; the
same comparisons run against the gfx950 build in tests/test_gpu_*.py."""
import pytest
import torch

from iplan_amd import _lib as L
from tests.emu.emu_lib import get_emu_lib


@pytest.fixture(autouse=True)
def emu():
    L.use_library_for_tests(get_emu_lib())
    yield
    L.use_library_for_tests(None)


def rel_err(a, b):
    a, b = a.double(), b.double()
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_gat_forward_emulated(golden, tag):
    from iplan_amd.config import default_args
    from iplan_amd.nova.GAT_Net import GAT_Net
    g = golden("gat_" + tag)
    args = default_args("highway", use_cuda=False, max_vehicle_num=g["N"])
    net = GAT_Net(g["D"], args)
    net.load_state_dict(g["params"])
    with torch.no_grad():
        out = net(g["obs"], g["h_prev"], noise=g["noise"])
    assert rel_err(out, g["out"]) < 1e-5
