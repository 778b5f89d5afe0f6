"""CPU (host-emulated kernel build): fused actor / critic backward vs autograd of the oracle."""
import pytest
import torch

from iplan_amd import _lib as L
from iplan_amd import ops, synth
from iplan_amd.config import default_args
from oracle import iplan_oracle as O
from tests.emu.emu_lib import get_emu_lib


@pytest.fixture(autouse=True)
def emu():
    L.use_library_for_tests(get_emu_lib())
    yield
    L.use_library_for_tests(None)


def rel_err(a, b):
    a, b = a.double(), b.double()
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


def test_actor_critic_backward_matches_autograd():
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    args = default_args("highway", use_cuda=False, max_vehicle_num=4, n_agents=2, episode_limit=5)
    torch.manual_seed(3)
    mac = DcntrlMAC(synth.make_scheme(args), {"agents": 2}, args)
    E, T, T1, nA, N = 4, 5, 6, 2, 4
    f = synth.make_episode_fields(args, E, seed=5, terminated_p=0.3)
    rows = E * T - 3                                     # ragged last tile, rows not a multiple of T is illegal -> use E*T
    rows = E * T
    srcs = []
    for key, w in (("history", 5), ("attention_latent", 32), ("behavior_latent", 8)):
        t = f[key]                                       # [E, T1, nA, N, w]
        srcs.append((t, w, t.stride(2), t.stride(1)))
    # last action: action of the previous step (training layout: dcntrl_controller.py:107 uses action[0] at t = 0)
    acts = f["actions"][..., 0]                          # [E, T1, nA]
    last = torch.cat([acts[:, :1], acts[:, :-1]], 1).to(torch.int32).contiguous()
    spec = ops.AcFeatureSpec(N, srcs, n_actions=5, last_action=last, la_strides=(1, nA), n_id=nA, T=T, T_phys=T1)
    ha, hc = f["rnn_states_actors"], f["rnn_states_critics"]          # [E, T1, nA, M]
    avail = f["avail_actions"]
    actions = f["actions"]
    out = ops.ac_forward(mac.actor_arena, mac.critic_arena, 2, spec, rows, nA, h_actor=ha, h_critic=hc,
                         h_strides=(ha.stride(2), ha.stride(1)), avail=avail, avail_strides=(avail.stride(2), avail.stride(1)),
                         mode=2, actions_in=actions, act_strides=(actions.stride(2), actions.stride(1)), n_actions=5,
                         ksplit=1, save=True, want_entropy=True, want_h=False)
    g_logp = torch.randn(nA, rows)
    g_v = torch.randn(nA, rows)
    g_ent = -0.01 / rows
    ops.ac_backward(out, mac.actor_arena, mac.critic_arena, g_logp=g_logp, g_entropy=g_ent, g_values=g_v)
    for i in range(nA):
        ap = {k: v.detach().clone().double().requires_grad_(v.requires_grad) for k, v in mac.agents[i].state_dict(keep_vars=True).items()}
        cp = {k: v.detach().clone().double().requires_grad_(v.requires_grad) for k, v in mac.critics[i].state_dict(keep_vars=True).items()}
        x = O.build_inputs_train(i, f["history"][:, :, i], f["attention_latent"][:, :, i], f["behavior_latent"][:, :, i],
                                 f["actions_onehot"][:, :, i], nA)[:, :-1].reshape(rows, -1).double()
        lp, _ = O.actor_evaluate(ap, x, ha[:, :-1, i].reshape(rows, -1).double(), actions[:, :-1, i].reshape(rows, 1),
                                 avail[:, :-1, i].reshape(rows, -1))
        logits, _ = O.actor_logits(ap, x, ha[:, :-1, i].reshape(rows, -1).double(), avail[:, :-1, i].reshape(rows, -1))
        la = torch.log_softmax(logits, -1)
        ent_rows = -(la.exp() * la).sum(-1)
        assert rel_err(out["logp"][i], lp[:, 0]) < 1e-5
        assert rel_err(out["entropy"][i], ent_rows) < 1e-5
        ((lp[:, 0] * g_logp[i].double()).sum() + g_ent * ent_rows.sum()).backward()
        v, _ = O.critic_value(cp, x, hc[:, :-1, i].reshape(rows, -1).double())
        assert rel_err(out["values"][i], v[:, 0]) < 1e-5
        (v[:, 0] * g_v[i].double()).sum().backward()
        for name, prm, arena in (("actor", ap, mac.actor_arena), ("critic", cp, mac.critic_arena)):
            for k in prm:
                got = arena.grad_of(i, k)
                ref = prm[k].grad if prm[k].grad is not None else torch.zeros_like(prm[k])
                err = (got.double() - ref).abs().max().item()
                assert err <= 1e-5 * ref.abs().max().item() + 1e-12, (name, i, k, err, ref.abs().max().item())


def test_module_level_autograd():
    """R_Actor.evaluate_actions / R_Critic.forward / GAT_Net.forward called as plain nn.Modules under
    autograd (the way the reference's learner calls them) give the oracle's gradients."""
    from iplan_amd.modules.agents.ippo_actor import R_Actor
    from iplan_amd.modules.critics.ippo_critic import R_Critic
    from iplan_amd.nova.GAT_Net import GAT_Net
    args = default_args("highway", use_cuda=False, max_vehicle_num=3, n_agents=2)
    torch.manual_seed(11)
    F, R = 40, 19
    actor, critic = R_Actor(F, args), R_Critic(F, args)
    x = torch.randn(R, 1, F)
    h = torch.randn(1, R, 64) * 0.1
    act = torch.randint(0, 5, (R, 1, 1))
    avail = torch.ones(R, 1, 5, dtype=torch.int32)
    avail[::3, 0, 2] = 0
    ap = {k: v.detach().clone().double().requires_grad_(v.requires_grad) for k, v in actor.state_dict(keep_vars=True).items()}
    cp = {k: v.detach().clone().double().requires_grad_(v.requires_grad) for k, v in critic.state_dict(keep_vars=True).items()}
    logp, ent = actor.evaluate_actions(x, h, act, avail)
    w = torch.randn(R, 1)
    ((logp * w).sum() - 0.3 * ent).backward()
    v, _ = critic(x, h)
    (v.reshape(-1) * w.reshape(-1)).sum().backward()
    lp, en = O.actor_evaluate(ap, x[:, 0].double(), h[0].double(), act.reshape(R, 1), avail.reshape(R, 5))
    ((lp * w.double()).sum() - 0.3 * en).backward()
    vv, _ = O.critic_value(cp, x[:, 0].double(), h[0].double())
    (vv[:, 0] * w.reshape(-1).double()).sum().backward()
    for mod, prm in ((actor, ap), (critic, cp)):
        for k, p in mod.named_parameters():
            if prm[k].grad is None:
                continue
            err = (p.grad.double() - prm[k].grad).abs().max().item()
            assert err <= 1e-5 * prm[k].grad.abs().max().item() + 1e-12, (k, err)
    # second backward accumulates (torch semantics)
    g1 = actor.base.mlp.fc1[0].bias.grad.clone()
    logp, ent = actor.evaluate_actions(x, h, act, avail)
    ((logp * w).sum() - 0.3 * ent).backward()
    assert torch.allclose(actor.base.mlp.fc1[0].bias.grad, 2 * g1, rtol=1e-5, atol=1e-8)
    # GAT module
    N, D, B = 3, 13, 2
    net = GAT_Net(D, args)
    obs, hp = torch.rand(B, N, D), torch.randn(B * N, 32) * 0.1
    noise = O.gumbel_noise_like_reference(B * N * (N - 1))
    gout = torch.randn(B * N, 32)
    out = net(obs, hp, noise=noise)
    (out * gout).sum().backward()
    gp = {k: v.detach().clone().double().requires_grad_(True) for k, v in net.state_dict().items()}
    o64 = O.gat_forward(gp, obs.double(), hp.double(), noise.double())
    (o64 * gout.double()).sum().backward()
    for k, p in net.named_parameters():
        err = (p.grad.double() - gp[k].grad).abs().max().item()
        assert err <= 1e-5 * gp[k].grad.abs().max().item() + 1e-10, (k, err)
