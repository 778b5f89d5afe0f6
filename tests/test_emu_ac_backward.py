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
