"""TEST INFRASTRUCTURE: the device-resident rollout that bench.py times (harness.SyntheticLoop._rollout_body: in-place
``write_back`` / ``out=`` launches on two streams) against the CPU oracle stepped the same way
(runners/ippo_parallel_runner.py:105-281 order of calls: select_actions_ippo -> [env.step] -> GAT_latent_update ->
latent_update -> EpisodeBatch.update), with the random draws (gumbel noise, exponential race) injected into both.
Shared by the emulated (CPU) and the ``-m gpu`` tests."""
import os

import torch

from oracle import iplan_oracle as O


def _cpu_sd(m):
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def oracle_rollout(loop, obs, noise, q_all):
    """Step the oracle through one episode of ``loop``'s networks on the pre-generated observation set ``obs``.
    noise [T+1, nA, E, N, N-1, 2], q_all [T, nA, E, n_actions] (CPU tensors).  Returns the expected episode fields
    [E, T+1, nA, ...] (attention_latent, behavior_latent, rnn_states_actors, rnn_states_critics, actions, actions_onehot)
    plus per-step values / logp."""
    a, E = loop.args, loop.E
    T, nA, N, L = a.episode_limit, a.n_agents, a.max_vehicle_num, a.max_history_len
    d, Z, A, M, R = a.obs_shape_single, a.latent_dim, a.attention_dim, a.rnn_hidden_dim, a.encoder_rnn_dim
    gat_on, beh_on = loop.prediction is not None, loop.behavior is not None
    hist_all = obs["hist"].cpu()                                          # [T1 + L - 1, E, nA, N, d]
    history = hist_all[L - 1:L + T].permute(1, 0, 2, 3, 4)                 # [E, T1, nA, N, d]
    gat_p = [_cpu_sd(m) for m in loop.prediction.pred_GAT] if gat_on else None
    enc_p = [_cpu_sd(m) for m in loop.behavior.behavior_encoder] if beh_on else None
    act_p = [_cpu_sd(m) for m in loop.mac.agents]
    cri_p = [_cpu_sd(m) for m in loop.mac.critics]
    T1 = T + 1
    att = torch.zeros(E, T1, nA, N, A)
    lat = torch.zeros(E, T1, nA, N, Z)
    ha = torch.zeros(E, T1, nA, M)
    hc = torch.zeros(E, T1, nA, M)
    actions = torch.zeros(E, T1, nA, 1, dtype=torch.long)
    onehot = torch.zeros(E, T1, nA, a.n_actions)
    values = torch.zeros(E, T, nA)
    logp = torch.zeros(E, T, nA)
    eh = torch.zeros(E, 1, nA, N, R)

    def gat_update(h_t, att_prev, lat_prev, nz):
        out = []
        for i in range(nA):
            x = torch.cat([h_t[:, i], lat_prev[:, i]], -1) if a.GAT_use_behavior else h_t[:, i]
            o = O.gat_forward(gat_p[i], x, att_prev[:, i].reshape(E * N, A), nz[i].reshape(-1, 2))
            out.append(o.reshape(E, N, A))
        return torch.stack(out, 1)

    with torch.no_grad():
        if gat_on:
            att[:, 0] = gat_update(history[:, 0], att[:, 0], lat[:, 0], noise[T])
        for t in range(T):
            last = onehot[:, t - 1] if t > 0 else torch.zeros_like(onehot[:, 0])
            x = O.build_inputs_rollout(history[:, t], att[:, t], lat[:, t], last, nA, gat_on, beh_on)
            for i in range(nA):
                logits, hn = O.actor_logits(act_p[i], x[:, i], ha[:, t, i], torch.ones(E, a.n_actions, dtype=torch.int32), use_relu=getattr(a, "use_ReLU", True))
                pr = torch.softmax(logits, -1)
                act = (pr / q_all[t, i]).argmax(-1)
                actions[:, t, i, 0] = act
                onehot[:, t, i] = torch.nn.functional.one_hot(act, a.n_actions).float()
                logp[:, t, i] = torch.log_softmax(logits, -1).gather(-1, act[:, None])[:, 0]
                ha[:, t + 1, i] = hn
                v, hcn = O.critic_value(cri_p[i], x[:, i], hc[:, t, i], use_relu=getattr(a, "use_ReLU", True))
                values[:, t, i] = v[:, 0]
                hc[:, t + 1, i] = hcn
            if gat_on:
                att[:, t + 1] = gat_update(history[:, t + 1], att[:, t], lat[:, t], noise[t])
            if beh_on:
                window = hist_all[t + 1:t + 1 + L].permute(1, 2, 3, 0, 4)          # [E, nA, N, L, d]
                lat[:, t + 1], eh = O.latent_update(enc_p, window, eh, lat[:, t], a.soft_update_coef)
    return dict(history=history, attention_latent=att, behavior_latent=lat, rnn_states_actors=ha, rnn_states_critics=hc,
                actions=actions, actions_onehot=onehot, values=values, logp=logp)


def check_rollout_body(args, E, device, seed=0, tol=1e-5):
    """SyntheticLoop._rollout_body with injected draws == the oracle, field by field, every step."""
    from iplan_amd.harness import SyntheticLoop
    from iplan_amd.nova.GAT_Net import gumbel_noise
    loop = SyntheticLoop(args, E, seed=seed, device=device)
    T, nA, N = args.episode_limit, args.n_agents, args.max_vehicle_num
    gen = torch.Generator().manual_seed(seed + 11)
    u = torch.rand(T + 1, nA, E, N, N - 1, 2, generator=gen).clamp_min(1e-20)
    noise = -torch.log((-torch.log(u)).clamp_min(1e-20))                    # gumbel samples (same law as F.gumbel_softmax's)
    q_all = -torch.log(torch.rand(T, nA, E, args.n_actions, generator=gen).clamp_min(1e-20))   # Exp(1)
    assert gumbel_noise((1, 1, 1, 2, 1, 2), device).shape == (1, 1, 1, 2, 1, 2)
    obs = loop.obs_sets[0]
    batch = loop.new_batch()
    with torch.no_grad():
        loop._rollout_body(obs, batch, noise=noise.to(device), q_all=q_all.to(device))
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    from iplan_amd import ops
    assert not ops.fused_sync_error(), "a fused rollout launch gave up waiting for its latent updates"
    # the one-launch vector step (latent updates + the next step's action selection, iplan_gat_enc_ac_fwd) against the two-launch
    # form: the same source in the same order -- actions bit-equal; the float fields to fp32 round-off (on the GPU the two kernels
    # are separate instantiations of the scene body and the compiler contracts a few multiply-adds differently: 1e-7 measured,
    # scripts/dev/fused_step_probe.py; the emulated build is bit-identical)
    if not os.environ.get("IPLAN_NO_FUSE_AC") and loop.prediction is not None and loop.behavior is not None:
        os.environ["IPLAN_NO_FUSE_AC"] = "1"
        try:
            batch2 = loop.new_batch()
            with torch.no_grad():
                loop._rollout_body(obs, batch2, noise=noise.to(device), q_all=q_all.to(device))
        finally:
            del os.environ["IPLAN_NO_FUSE_AC"]
        for k in ("actions", "actions_onehot"):
            assert torch.equal(batch[k], batch2[k]), ("fused vs two-launch step differ", k)
        for k in ("attention_latent", "behavior_latent", "rnn_states_actors", "rnn_states_critics"):
            e = (batch[k].double() - batch2[k].double()).abs().max().item()
            assert e <= 2e-6 * max(1.0, batch2[k].abs().max().item()), ("fused vs two-launch step differ", k, e)
    ref = oracle_rollout(loop, obs, noise, q_all)
    got = {k: batch[k].cpu() for k in ("history", "attention_latent", "behavior_latent", "rnn_states_actors", "rnn_states_critics",
                                       "actions", "actions_onehot")}
    assert torch.equal(got["actions"][:, :T], ref["actions"][:, :T]), "actions differ"
    assert torch.equal(got["actions_onehot"][:, :T], ref["actions_onehot"][:, :T])
    worst = {}
    for k in ("history", "attention_latent", "behavior_latent", "rnn_states_actors", "rnn_states_critics"):
        for t in range(T + 1):
            e = (got[k][:, t].double() - ref[k][:, t].double()).abs().max().item() / max(1.0, ref[k][:, t].abs().max().item())
            worst[k] = max(worst.get(k, 0.0), e)
            assert e < tol, (k, t, e)
    return worst
