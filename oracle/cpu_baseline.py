"""TEST / MEASUREMENT INFRASTRUCTURE: the bounded CPU sample bench.py reports as ``cpu_baseline`` -- the same four pieces of
one training cycle timed either on the oracle (``backend="oracle"``: the in-repo restatement, what the GPU box can run) or
on the REAL reference classes imported from /root/reference (``backend="reference"``: build container only).
``scripts/anchor_cpu_baseline.py`` runs both on identical inputs and commits the ratio (profiles/r0N_cpu_baseline_anchor.json),
which ties the oracle's speed to the reference's (round 6: profiles/r06_cpu_baseline_anchor.json, this round's oracle -- behaviour leg of all
agents at Eb = 32, PPO leg on all 22 950 rows; round 3's anchor: profiles/r03_cpu_baseline_anchor.json).

Pieces (each: one warm-up call, then the best of ``reps`` timed calls), scaled to seconds per env-step and summed:
  rollout   vector step at full width E: GAT_latent_update + latent_update + select_actions_ippo (5 agents)
  behaviour Behavior_policy.learn forward + backward, ALL n_agents agents one after the other (the reference's loop), Eb envs,
            full 90-step episode -- the dominant CPU leg: timed in full at the width the GPU run uses, not extrapolated from one agent
  predict   Prediction_policy.learn forward + backward, ONE agent, 64 samples
  ppo       one PPO epoch (actor evaluate + critic, forward + backward), ONE agent, all Rp = batch_size x episode_limit rows
"""
import os
import sys
import time

import torch


def _best(fn, reps):
    fn()                                                   # warm-up (allocator, thread pool, first-call dispatch)
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def _oracle_pieces(ca, E, Eb, Rp):
    from iplan_amd import synth
    from iplan_amd.modules.agents.ippo_actor import R_Actor
    from iplan_amd.modules.critics.ippo_critic import R_Critic
    from iplan_amd.nova.GAT_Net import GAT_Net
    from iplan_amd.nova.behavior_net import Behavior_Latent_Decoder, EncoderRNN
    from iplan_amd.nova.prediction_net import Prediction_Decoder
    from oracle import iplan_oracle as O
    nA, N, d, Z, A, L = ca.n_agents, ca.max_vehicle_num, ca.obs_shape_single, ca.latent_dim, ca.attention_dim, ca.max_history_len
    sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}  # noqa: E731
    req = lambda p: {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in p.items()}  # noqa: E731
    gat = [sd(GAT_Net(d + Z, ca)) for _ in range(nA)]
    enc = [sd(EncoderRNN(d, 32, Z, 1)) for _ in range(nA)]
    bdec = sd(Behavior_Latent_Decoder(d + Z, 64, 1, d, 0.1))
    pdec = sd(Prediction_Decoder(d, A, 1, d, ca.pred_length, 0.1, 0))
    F = N * (d + A + Z) + ca.n_actions + nA
    act = [sd(R_Actor(F, ca)) for _ in range(nA)]
    cri = [sd(R_Critic(F, ca)) for _ in range(nA)]
    hist, window = synth.rollout_step_inputs(ca, E, 0)
    hist, window = torch.as_tensor(hist, dtype=torch.float32), torch.as_tensor(window, dtype=torch.float32)
    st = dict(att=torch.zeros(E, nA, N, A), lat=torch.full((E, nA, N, Z), 1.0 / Z), eh=torch.zeros(E, 1, nA, N, 32))
    ha = torch.zeros(E, nA, 64)

    def rollout():
        with torch.no_grad():
            new_att = []
            for i in range(nA):
                noise = O.gumbel_noise_like_reference(E * N * (N - 1))
                new_att.append(O.gat_forward(gat[i], torch.cat([hist[:, i], st["lat"][:, i]], -1),
                                             st["att"][:, i].reshape(E * N, A), noise).reshape(E, N, A))
            st["att"] = torch.stack(new_att, 1)
            st["lat"], st["eh"] = O.latent_update(enc, window, st["eh"], st["lat"], ca.soft_update_coef)
            x = O.build_inputs_rollout(hist, st["att"], st["lat"], torch.zeros(E, nA, ca.n_actions), nA)
            for i in range(nA):
                O.actor_logits(act[i], x[:, i], ha[:, i])
                O.critic_value(cri[i], x[:, i], ha[:, i])
    f = synth.make_episode_fields(ca, Eb, seed=1, terminated_p=0.5)

    bdecs = [bdec] + [sd(Behavior_Latent_Decoder(d + Z, 64, 1, d, 0.1)) for _ in range(nA - 1)]

    def behaviour(agents=range(nA)):
        for i in agents:
            ep, dp = req(enc[i]), req(bdecs[i])
            _, _, loss = O.behavior_learn_loss(ep, dp, f["history"][:, :-1, i], f["terminated"][:, :-1, i, 0].float(), L,
                                               ca.soft_update_coef, None, 0.0)
            loss.backward()
    S = ca.pred_batch_size
    gen = torch.Generator().manual_seed(2)
    obs = synth.make_history(gen, (S,), N, d)
    lat = torch.softmax(torch.randn(S, N, Z, generator=gen), -1)
    att = torch.randn(S, N, 1, A, generator=gen) * 0.1
    actual = synth.make_history(gen, (S, N), ca.pred_length, d)

    def predict():
        gp, dp = req(gat[0]), req(pdec)
        noise = O.gumbel_noise_like_reference(S * N * (N - 1))
        loss, _ = O.prediction_loss(gp, dp, obs.unsqueeze(2), att, lat.unsqueeze(2), actual, torch.ones_like(actual), noise,
                                    None, 0.0, ca.pred_length)
        loss.backward()
    xr, hr = torch.randn(Rp, F), torch.randn(Rp, 64) * 0.1
    ar = torch.randint(0, ca.n_actions, (Rp, 1))

    def ppo():
        ap, cp = req(act[0]), req(cri[0])
        lp, ent = O.actor_evaluate(ap, xr, hr, ar)
        v, _ = O.critic_value(cp, xr, hr)
        (lp.sum() + ent).backward()
        v.sum().backward()
    return dict(rollout=rollout, behaviour=behaviour, predict=predict, ppo=ppo)


def _reference_pieces(ca, E, Eb, Rp, ref_root):
    """The same four pieces on the reference's own classes (its Python loops, numpy round trips and autograd graphs included)."""
    import copy
    import numpy as np
    sys.path.insert(0, ref_root)
    from components.episode_buffer import EpisodeBatch
    from components.transforms import OneHot
    from controllers.dcntrl_controller import DcntrlMAC
    from modules.agents.ippo_actor import R_Actor
    from modules.critics.ippo_critic import R_Critic
    from nova.prediction_policy import Prediction_policy
    from nova.stable_behavior_policy import Behavior_policy
    from iplan_amd import synth

    class Log:
        def log_stat(self, *a, **k):
            pass
    nA, N, d, Z, A = ca.n_agents, ca.max_vehicle_num, ca.obs_shape_single, ca.latent_dim, ca.attention_dim
    ca = copy.copy(ca)
    ca.obs_shape, ca.state_shape = d * 5, d * 5
    scheme = synth.make_scheme(ca)
    pred, beh = Prediction_policy(ca, Log()), Behavior_policy(ca, Log())
    mac = DcntrlMAC(scheme, {"agents": nA}, ca)

    def episode_batch(args, E_, seed):
        sch = synth.make_scheme(args)
        sch.pop("actions_onehot")
        sch.pop("filled")
        b = EpisodeBatch(sch, {"agents": args.n_agents}, E_, args.episode_limit + 1,
                         preprocess={"actions": ("actions_onehot", [OneHot(out_dim=args.n_actions)])}, device="cpu")
        for k, v in synth.make_episode_fields(args, E_, seed, 0.5).items():
            b.data.transition_data[k].copy_(v.view_as(b.data.transition_data[k]))
        return b
    hist, window = synth.rollout_step_inputs(ca, E, 0)
    st = dict(att=np.zeros((E, nA, N, A), dtype=np.float32), lat=np.full((E, nA, N, Z), 1.0 / Z, dtype=np.float32),
              eh=np.zeros((E, 1, nA, N, 32), dtype=np.float32))
    roll_batch = episode_batch(ca, E, 3)

    def rollout():
        st["att"] = pred.GAT_latent_update(hist, st["att"], st["lat"])
        st["lat"], st["eh"] = beh.latent_update(window, st["eh"], st["lat"])
        mac.select_actions_ippo(roll_batch, t_ep=1)
    one = copy.copy(ca)
    one.n_agents = 1
    one.batch_size_run = Eb
    pred1 = Prediction_policy(one, Log())
    b64 = episode_batch(one, max(Eb, 2), 2)
    allb = copy.copy(ca)
    allb.batch_size_run = Eb
    beh_all = Behavior_policy(allb, Log())                   # learn() loops over all n_agents agents itself
    b_all = episode_batch(allb, Eb, 1)
    beh1 = Behavior_policy(one, Log())
    b1 = episode_batch(one, Eb, 1)

    def behaviour(agents=None):
        (beh_all.learn(b_all, 0) if agents is None else beh1.learn(b1, 0))

    def predict():
        pred1.learn(b64, 0)
    F = N * (d + A + Z) + ca.n_actions + nA
    actor, critic = R_Actor(F, ca), R_Critic(F, ca)
    xr, hr = torch.randn(Rp, 1, F), torch.randn(1, Rp, 64) * 0.1
    ar, av = torch.randint(0, ca.n_actions, (Rp, 1, 1)), torch.ones(Rp, 1, ca.n_actions)

    def ppo():
        actor.zero_grad()
        critic.zero_grad()
        lp, ent = actor.evaluate_actions(xr, hr, ar, av)
        v, _ = critic(xr, hr)
        (lp.sum() + ent).backward()
        v.sum().backward()
    return dict(rollout=rollout, behaviour=behaviour, predict=predict, ppo=ppo)


def measure(backend, E, cores, Eb=2, Rp=None, reps=3, ref_root="/root/reference"):
    """-> dict(value env-steps/s, seconds per piece call, per_env_step seconds per piece, description).
    ``Rp``: rows of the PPO leg; default = ALL batch_size x episode_limit rows one agent trains on per epoch (22 950 at config 3;
    until round 5 a 2 048-row sample scaled linearly -- VERDICT r5 "weak" 4)."""
    from iplan_amd.config import default_args
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    ca = default_args("highway", use_cuda=False)
    if Rp is None:
        Rp = ca.batch_size * ca.episode_limit
    pieces = _oracle_pieces(ca, E, Eb, Rp) if backend == "oracle" else _reference_pieces(ca, E, Eb, Rp, ref_root)
    t = {k: _best(fn, reps if k != "rollout" else max(reps, 5)) for k, fn in pieces.items() if k != "behaviour"}
    # the behaviour leg is the long one (all agents, full width): warm up on ONE agent, then time the whole pass once
    pieces["behaviour"]([0] if backend == "oracle" else 1)
    t0 = time.perf_counter()
    pieces["behaviour"]()
    t["behaviour"] = time.perf_counter() - t0
    nA, T = ca.n_agents, ca.episode_limit
    per = dict(rollout=t["rollout"] / E,                                             # one vector step serves E env-steps
               behaviour=t["behaviour"] / (Eb * T),                                  # Behavior_policy.learn (all agents) once per rollout
               predict=nA * t["predict"] / (E * T),                                  # Prediction_policy.learn once per rollout
               ppo=nA * ca.ppo_epoch * t["ppo"] / Rp * (ca.batch_size / ca.buffer_size))     # 15 epochs over 255/256 of the rows
    return dict(value=1.0 / sum(per.values()), seconds=t, per_env_step=per, backend=backend, cores=cores,
                sample=f"{backend} on {cores} threads, best of {reps} after a warm-up call: rollout vector step at E={E} ({t['rollout']:.3f}s); "
                       f"Behaviour learn fwd+bwd ALL {nA} agents x {Eb} envs x full episode, timed once after a one-agent warm-up ({t['behaviour']:.2f}s); Prediction learn fwd+bwd 1 agent x "
                       f"{ca.pred_batch_size} samples ({t['predict']:.2f}s); one PPO epoch 1 agent x {Rp} rows ({t['ppo']:.3f}s); each scaled "
                       "linearly to s/env-step (the one-agent legs x n_agents) and summed")
