"""GPU parity: fused GAT kernel on a real MI355X vs the committed reference outputs and the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


@pytest.mark.parametrize("tag", ["small", "wide", "hwy"])
def test_gat_forward_matches_reference(golden, tag):
    from iplan_amd.config import default_args
    from iplan_amd.nova.GAT_Net import GAT_Net
    g = golden("gat_" + tag)
    args = default_args("highway", use_cuda=True, max_vehicle_num=g["N"])
    net = GAT_Net(g["D"], args)
    net.load_state_dict(g["params"])
    with torch.no_grad():
        out = net(g["obs"].cuda(), g["h_prev"].cuda(), noise=g["noise"].cuda())
    assert rel_err(out, g["out"]) < 1e-5      # tolerance: BASELINE.json north_star (1e-5 fp32)


def test_gat_forward_stacked_nets_vs_oracle():
    """cfg3-shaped launch (5 nets x 8 envs x 55 entities, strided env-major inputs) vs the oracle."""
    from iplan_amd import ops
    from iplan_amd.arena import ParamArena
    from iplan_amd.config import default_args
    from iplan_amd.nova.GAT_Net import GAT_Net, gumbel_noise
    from oracle import iplan_oracle as O
    args = default_args("highway", use_cuda=True)
    torch.manual_seed(0)
    nA, E, N, d, Z, A = 5, 8, 55, 5, 8, 32
    nets = [GAT_Net(d + Z, args) for _ in range(nA)]
    cpu_params = [{k: v.detach().clone() for k, v in n.state_dict().items()} for n in nets]
    arena = ParamArena(nets, "cuda")
    hist = torch.rand(E, nA, N, d) * 2 - 1
    lat = torch.softmax(torch.randn(E, nA, N, Z), -1)
    hid = torch.randn(E, nA, N, A) * 0.1
    noise = gumbel_noise((nA, E, N, N - 1, 2), "cpu")
    out, _ = ops.gat_forward(arena, hist.cuda().permute(1, 0, 2, 3), lat.cuda().permute(1, 0, 2, 3),
                             hid.cuda().permute(1, 0, 2, 3), noise.cuda())
    for i in range(nA):
        ref = O.gat_forward(cpu_params[i], torch.cat([hist[:, i], lat[:, i]], -1), hid[:, i].reshape(E * N, A),
                            noise[i].reshape(-1, 2))
        assert rel_err(out[i].reshape(E * N, A), ref) < 1e-5, i


def gat_vs_oracle(B, N, D, seed, device="cuda"):
    """One GAT_Net forward of random weights / inputs against the CPU oracle (any entity count 2..64)."""
    from iplan_amd.config import default_args
    from iplan_amd.nova.GAT_Net import GAT_Net, gumbel_noise
    from oracle import iplan_oracle as O
    args = default_args("highway", use_cuda=(device != "cpu"))
    torch.manual_seed(seed)
    net = GAT_Net(D, args)
    params = {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}
    obs = torch.rand(B, N, D) * 2 - 1
    h_prev = torch.randn(B * N, args.attention_dim) * 0.1
    noise = gumbel_noise((B * N * (N - 1), 2), "cpu")
    with torch.no_grad():
        out = net(obs.to(device), h_prev.to(device), noise=noise.to(device))
    ref = O.gat_forward(params, obs, h_prev, noise)
    assert rel_err(out, ref) < 1e-5, rel_err(out, ref)
