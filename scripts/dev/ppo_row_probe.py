"""Debug probe 2: per-ROW comparison of the actor's stored activations / row gradients of the last PPO epoch against an fp64
autograd evaluation at the SAME parameters (the arenas as they were when that epoch's backward ran)."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from iplan_amd import ops, synth
from iplan_amd.config import default_args
from tests.oracle_checks import _fields, _Log, _sd, _req
from oracle import iplan_oracle as O

def main():
    args = default_args("highway", use_cuda=True, ppo_epoch=2)
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    from iplan_amd.learners.ippo_learner import IPPOLearner
    torch.manual_seed(24)
    scheme = synth.make_scheme(args)
    mac = DcntrlMAC(scheme, {"agents": args.n_agents}, args)
    pre = dict(actors=[_sd(m) for m in mac.agents], critics=[_sd(m) for m in mac.critics])
    learner = IPPOLearner(mac, scheme, _Log(), args)
    E = args.buffer_size
    fields, batch = _fields(args, E, 25, 0.15, "cuda")
    learner.batch_size_run = E
    learner.insert_episode_batch(batch)
    rec = []
    orig = ops.ac_backward
    def spy(fwd, *a, **k):
        out = orig(fwd, *a, **k)
        if os.environ.get("PROBE_SYNC"):
            torch.cuda.synchronize()
        rec.append(dict(saved=fwd["saved"][0, 0].clone(), dsave=out["dsave"][0, 0].clone(), aparam=mac.actor_arena.data[0].clone(),
                        g_logp=k["g_logp"][0].clone(), logp=fwd["logp"][0].clone(), agrad=mac.actor_arena.grad[0].clone()))
        return out
    ops.ac_backward = spy
    learner.probe_last_step = True
    learner.train(0)
    torch.cuda.synchronize()
    print("last_step_params vs spy params (actor, critic):", float((learner.last_step_params[0][0] - rec[-1]["aparam"]).abs().max()))
    i, M = 0, 64
    arena = mac.actor_arena
    f64 = {k: (v.double() if v.is_floating_point() else v) for k, v in fields.items()}
    a1e = copy.copy(args); a1e.ppo_epoch = 1
    ap, cp = _req(pre["actors"][i], torch.float64), _req(pre["critics"][i], torch.float64)
    r0 = O.ppo_train_agent(i, ap, cp, f64, a1e)                   # old_logp / adv / returns of the pre-train parameters
    T = args.episode_limit
    rows = args.batch_size * T
    x_all = O.build_inputs_train(i, f64["history"][:, :, i], f64["attention_latent"][:, :, i], f64["behavior_latent"][:, :, i],
                                 f64["actions_onehot"][:, :, i], args.n_agents, True, True)
    x = x_all[:, :-1].reshape(-1, x_all.shape[-1])[:rows]
    ha = f64["rnn_states_actors"][:, :-1, i].reshape(-1, M)[:rows]
    acts = f64["actions"][:, :-1, i].reshape(-1, 1)[:rows]
    avail = f64["avail_actions"][:, :-1, i].reshape(-1, args.n_actions)[:rows]
    masks = (1.0 - f64["terminated"][:, :-1, i].double()).reshape(-1, 1)[:rows]
    from tests.oracle_checks import probe_dicts
    probe = probe_dicts(learner, mac, pre, i)
    ap3, cp3 = _req(pre["actors"][i], torch.float64), _req(pre["critics"][i], torch.float64)
    r3 = O.ppo_train_agent(i, ap3, cp3, f64, args, probe_last_step=probe)
    for k in ("base.mlp.fc2.0.0.weight", "base.mlp.fc1.0.bias", "rnn.rnn.weight_ih_l0"):
        n = int(torch.Size(arena.shapes[k]).numel())
        g = rec[-1]["agrad"][arena.offsets[k]:arena.offsets[k] + n].view(arena.shapes[k]).double().cpu()
        pg = r3["probe_grads"][0][k]
        print(f"  oracle probe path {k}: vs arena {float((g - pg).abs().max() / pg.abs().max()):.2e}; probe param vs spy param "
              f"{float((probe[0][k].double() - rec[-1]['aparam'][arena.offsets[k]:arena.offsets[k] + n].view(arena.shapes[k]).double().cpu()).abs().max()):.2e}")
    globals()["_r3"] = r3
    if os.environ.get("IPLAN_DUMP"):
        torch.save(dict(probe=probe, g64=[{k: v.clone() for k, v in g.items()} for g in r3["probe_grads"]],
                        kernel={k: mac.actor_arena.grad_of(i, k).detach().cpu().clone() for k in mac.actor_arena.names},
                        old_logp=r3["old_logp"], adv=r3["adv"]), os.environ["IPLAN_DUMP"])
        return
    for ep, r in enumerate(rec):
        def P(k):
            n = int(torch.Size(arena.shapes[k]).numel())
            return r["aparam"][arena.offsets[k]:arena.offsets[k] + n].view(arena.shapes[k]).double().cpu().requires_grad_(True)
        p = {k: P(k) for k in arena.names}
        ln = lambda v, w, b: (v - v.mean(-1, keepdim=True)) / torch.sqrt(((v - v.mean(-1, keepdim=True)) ** 2).mean(-1, keepdim=True) + 1e-5) * w + b
        xn = ln(x, p["base.feature_norm.weight"], p["base.feature_norm.bias"])
        z1 = xn @ p["base.mlp.fc1.0.weight"].t() + p["base.mlp.fc1.0.bias"]; z1.retain_grad()
        a1 = torch.relu(z1)
        f1 = ln(a1, p["base.mlp.fc1.2.weight"], p["base.mlp.fc1.2.bias"])
        z2 = f1 @ p["base.mlp.fc2.0.0.weight"].t() + p["base.mlp.fc2.0.0.bias"]; z2.retain_grad()
        a2 = torch.relu(z2)
        f2 = ln(a2, p["base.mlp.fc2.0.2.weight"], p["base.mlp.fc2.0.2.bias"]); f2.retain_grad()
        hn = O.gru_cell(f2, ha, p["rnn.rnn.weight_ih_l0"], p["rnn.rnn.weight_hh_l0"], p["rnn.rnn.bias_ih_l0"], p["rnn.rnn.bias_hh_l0"])
        f3 = ln(hn, p["rnn.norm.weight"], p["rnn.norm.bias"])
        logits = f3 @ p["act.action_out.linear.weight"].t() + p["act.action_out.linear.bias"]
        logits = torch.where(avail == 0, torch.full_like(logits, -1e10), logits)
        lpa = torch.log_softmax(logits, -1)
        logp = lpa.gather(-1, acts.long()); logp.retain_grad()
        ent = -(lpa.exp() * lpa.clamp(min=torch.finfo(lpa.dtype).min)).sum(-1).mean()
        zero = torch.zeros(rows, 1, dtype=torch.float64)
        a_obj = O.ppo_losses(logp, ent, zero, r0["old_logp"][:rows], r0["adv"].reshape(-1, 1)[:rows], zero, zero, masks,
                             args.clip_param, args.huber_delta, args.entropy_coef, args.value_loss_coef)[0]
        a_obj.backward()
        sv, ds = r["saved"].double().cpu(), r["dsave"].double().cpu()
        def stat(name, got, ref):
            d = got - ref
            print(f"  ep{ep} {name:6s} max|ref|={float(ref.abs().max()):.3e} max|d|/max|ref|={float(d.abs().max() / ref.abs().max()):.2e} "
                  f"|sum_rows d|max={float(d.sum(0).abs().max()):.3e} |sum_rows ref|max={float(ref.sum(0).abs().max()):.3e} "
                  f"sum_rows|ref| max={float(ref.abs().sum(0).max()):.3e}")
        stat("a1", sv[:, 0:M], a1.detach()); stat("f1", sv[:, M:2*M], f1.detach()); stat("a2", sv[:, 2*M:3*M], a2.detach())
        stat("f2", sv[:, 3*M:4*M], f2.detach())
        stat("logp", r["logp"].double().cpu().reshape(-1, 1), logp.detach())
        stat("g_logp", r["g_logp"].double().cpu().reshape(-1, 1), logp.grad)
        stat("dz2", ds[:, M:2*M], z2.grad); stat("dz1", ds[:, 0:M], z1.grad)
        gi = torch.autograd.grad  # noqa
        # mask flips
        print(f"  ep{ep} relu mask flips: a1 {int(((sv[:, 0:M] > 0) != (a1.detach() > 0)).sum())}  a2 {int(((sv[:, 2*M:3*M] > 0) != (a2.detach() > 0)).sum())}")
        for k in ("base.mlp.fc2.0.0.weight", "base.mlp.fc2.0.0.bias", "base.mlp.fc1.0.bias", "rnn.rnn.weight_ih_l0"):
            n = int(torch.Size(arena.shapes[k]).numel())
            g = r["agrad"][arena.offsets[k]:arena.offsets[k] + n].view(arena.shapes[k]).double().cpu()
            print(f"  ep{ep} grad {k}: {float((g - p[k].grad).abs().max() / p[k].grad.abs().max()):.2e} (pre-clip autograd vs arena)")
            if ep == len(rec) - 1:
                pg = globals()["_r3"]["probe_grads"][0][k]
                print(f"      independent autograd vs oracle probe path: {float((pg - p[k].grad).abs().max() / pg.abs().max()):.2e}")
        if ep == len(rec) - 1:
            print("   old_logp r0 vs r3:", float((r0["old_logp"] - globals()["_r3"]["old_logp"]).abs().max()), " adv:", float((r0["adv"] - globals()["_r3"]["adv"]).abs().max()))

main()
