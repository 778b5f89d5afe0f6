"""CPU oracle for the iPLAN hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch, vectorised *restatement* (torch-CPU, dtype-generic: run it in
fp32 or fp64) of the arithmetic the reference performs on the north-star path.  It is NOT part of
the product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product path (``iplan_amd``) never imports it and fails loudly without the HIP
library.

Parity status: PINNED.  The reference ships no tests or golden vectors for this path (SURVEY.md
§4), so the pin is (i) the reference itself imported in the build container
(``oracle/make_golden.py`` runs the reference classes from /root/reference on seeded inputs and
asserts this restatement reproduces them) and (ii) the fixtures that script commits under
``tests/golden/`` which are re-checked on every test run, including on the GPU box where the
reference does not exist.

Every function cites the reference file:line it restates (paths relative to the reference root).
Parameters are passed as dicts keyed by the reference's ``state_dict`` names.
"""
import math

import torch

EPS = 1e-10  # nova/prediction_policy.py:12, nova/stable_behavior_policy.py:11


# ----------------------------------------------------------------------------------------------
# GRU primitives (torch.nn.GRU / GRUCell semantics: gate order r,z,n; b_hn inside the r* term)
# ----------------------------------------------------------------------------------------------
def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """One GRU step.  x [R,In], h [R,H] -> h' [R,H]."""
    H = h.shape[-1]
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1.0 - z) * n + z * h


def gru_seq(x, h0, w_ih, w_hh, b_ih, b_hh):
    """batch_first GRU over T steps.  x [R,T,In], h0 [R,H] -> (out [R,T,H], hT [R,H])."""
    h = h0
    outs = []
    for t in range(x.shape[1]):
        h = gru_cell(x[:, t], h, w_ih, w_hh, b_ih, b_hh)
        outs.append(h)
    return torch.stack(outs, dim=1), h


def neighbour_index(N, device=None):
    """j(i,s) = s + [s >= i]: the s-th *other* entity seen by ego i (nova/GAT_Net.py:58-69)."""
    i = torch.arange(N, device=device)[:, None]
    s = torch.arange(N - 1, device=device)[None, :]
    return s + (s >= i).long()  # [N, N-1]


def gumbel_noise_like_reference(n_rows, dtype=torch.float32):
    """Draw the noise exactly as F.gumbel_softmax does on a [n_rows,2] logits tensor
    (torch/nn/functional.py: ``-empty_like(logits).exponential_().log()``), consuming the global
    CPU generator the same way the reference does at nova/GAT_Net.py:93."""
    return -torch.empty(n_rows, 2, dtype=dtype).exponential_().log()


# ----------------------------------------------------------------------------------------------
# a1  GAT_Net.forward  (nova/GAT_Net.py:41-142)
# ----------------------------------------------------------------------------------------------
def gat_forward(p, obs, h_prev, noise, tau=0.01, return_internals=False):
    """obs [B,N,D], h_prev [B*N,A], noise [B*N*(N-1),2] (gumbel samples) -> [B*N,A].

    Uses the separable input projection W_ih[h_i;h_j] = W_a h_i + W_b h_j instead
    of materialising the [N-1, B*N, 2H] pair tensor of nova/GAT_Net.py:57-75.
    """
    B, N, _ = obs.shape
    H = p["encoding.weight"].shape[0]
    A = p["q.weight"].shape[0]
    h = torch.relu(obs @ p["encoding.weight"].t() + p["encoding.bias"])  # :49   [B,N,H]
    jidx = neighbour_index(N, obs.device)                               # [N,N-1]

    outs = []
    for sfx, order in (("", range(N - 1)), ("_reverse", range(N - 2, -1, -1))):   # :78-83
        w_ih = p["hard_bi_GRU.weight_ih_l0" + sfx]
        w_hh = p["hard_bi_GRU.weight_hh_l0" + sfx]
        b_ih = p["hard_bi_GRU.bias_ih_l0" + sfx]
        b_hh = p["hard_bi_GRU.bias_hh_l0" + sfx]
        a_proj = h @ w_ih[:, :H].t() + b_ih       # ego part,   [B,N,3H]
        b_proj = h @ w_ih[:, H:].t()              # other part, [B,N,3H]
        state = torch.zeros(B, N, H, dtype=obs.dtype, device=obs.device)
        out = [None] * (N - 1)
        for s in order:
            gi = a_proj + b_proj[:, jidx[:, s]]
            gh = state @ w_hh.t() + b_hh
            r = torch.sigmoid(gi[..., :H] + gh[..., :H])
            z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
            n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
            state = (1.0 - z) * n + z * state
            out[s] = state
        outs.append(torch.stack(out, dim=2))      # [B,N,N-1,H]
    hh = torch.cat(outs, dim=-1)                   # [B,N,N-1,2H]
    logits = hh @ p["hard_encoding.weight"].t() + p["hard_encoding.bias"]        # :91
    y = (logits + noise.reshape(B, N, N - 1, 2)) / tau                            # :93
    hard = torch.softmax(y, dim=-1)[..., 1]                                       # :95  [B,N,N-1]

    q = h @ p["q.weight"].t()                                                     # :101
    k = h @ p["k.weight"].t()                                                     # :103
    v = torch.relu(h @ p["v.weight"].t() + p["v.bias"])                           # :105
    kj = k[:, jidx]                                                               # [B,N,N-1,A]
    vj = v[:, jidx]
    score = (q[:, :, None, :] * kj).sum(-1) / math.sqrt(A)                        # :123-126
    soft = torch.softmax(score, dim=-1)                                           # :129
    x = (vj * (soft * hard)[..., None]).sum(2)                                    # :132 (no renorm)
    out = gru_cell(x.reshape(B * N, A), h_prev, p["rnn.weight_ih"], p["rnn.weight_hh"],
                   p["rnn.bias_ih"], p["rnn.bias_hh"])                            # :140
    if return_internals:
        return out, dict(h=h, logits=logits, hard=hard, soft=soft, x=x)
    return out


# ----------------------------------------------------------------------------------------------
# a6  EncoderRNN.forward (nova/behavior_net.py:17-22)
# ----------------------------------------------------------------------------------------------
def encoder_forward(p, x, h0):
    """x [R,L,d], h0 [R,Rdim] -> (seq_out [R,L,Rdim], hL [R,Rdim], latent [R,Z] softmax)."""
    u = torch.relu(x @ p["linear.weight"].t() + p["linear.bias"])
    out, hL = gru_seq(u, h0, p["rnn.weight_ih_l0"], p["rnn.weight_hh_l0"],
                      p["rnn.bias_ih_l0"], p["rnn.bias_hh_l0"])
    latent = torch.softmax(hL @ p["out.weight"].t() + p["out.bias"], dim=-1)
    return out, hL, latent


# ----------------------------------------------------------------------------------------------
# DecoderRNN.forward (nova/behavior_net.py:39-45 and nova/prediction_net.py:20-26)
# ----------------------------------------------------------------------------------------------
def decoder_forward(p, x, h0, drop_mask=None, drop_p=0.0):
    """x [R,T,In], h0 [R,Hd], drop_mask [R,T,Hd] of {0,1} keep flags (None = no dropout)
    -> (y [R,T,Out], hT)."""
    u = torch.relu(x @ p["linear.weight"].t() + p["linear.bias"])
    out, hT = gru_seq(u, h0, p["rnn.weight_ih_l0"], p["rnn.weight_hh_l0"],
                      p["rnn.bias_ih_l0"], p["rnn.bias_hh_l0"])
    a = torch.tanh(out)
    if drop_mask is not None:
        a = a * drop_mask / (1.0 - drop_p)
    return a @ p["out.weight"].t() + p["out.bias"], hT


def strip_prefix(p, prefix):
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------------------------
# a5  Prediction_Decoder.forward (nova/prediction_net.py:40-63), teacher_forcing_ratio == 0
# ----------------------------------------------------------------------------------------------
def prediction_decoder_forward(p, last_state, hidden, pred_length, drop_masks=None, drop_p=0.0):
    """last_state [B,N,1,d], hidden [B*N,A] (GAT output) -> predicted [B,N,pred_length,d].
    drop_masks: [pred_length, B*N, 1, A] keep flags."""
    B, N, _, d = last_state.shape
    dp = strip_prefix(p, "decoder.")
    x = last_state.reshape(B * N, 1, d)
    h = hidden
    preds = []
    for t in range(pred_length):
        y, h = decoder_forward(dp, x, h, None if drop_masks is None else drop_masks[t], drop_p)
        preds.append(y)
        x = y
    return torch.cat(preds, dim=1).reshape(B, N, pred_length, d)


def masked_l1(target, pred, mask, scale):
    """sum(|t-p|*m)/(sum(m)+EPS)*scale  (nova/prediction_policy.py:223-225,
    nova/stable_behavior_policy.py:233-235)."""
    return (torch.abs(target - pred) * mask).sum() / (mask.sum() + EPS) * scale


# ----------------------------------------------------------------------------------------------
# a3/a4 Prediction_policy: batch assembly + loss (nova/prediction_policy.py:123-164, 168-225)
# ----------------------------------------------------------------------------------------------
def prediction_gather(history, attention, latent, mask, select_idx, pred_length):
    """history [E,T,N,d], attention [E,T,N,A], latent [E,T,N,Z], mask [E,T]; select_idx = the
    np.random.choice draw.  Returns (input_traj [S,N,1,d], input_att [S,N,1,A],
    input_lat [S,N,1,Z], actual [S,N,P,d], mask_over [S,N,P,d])."""
    E, T, N, d = history.shape
    avail = T - pred_length - 1
    sel = torch.as_tensor(select_idx, dtype=torch.long)
    b = sel // avail
    t = sel % avail
    input_traj = history[b, t].unsqueeze(2)
    input_att = attention[b, t].unsqueeze(2)
    input_lat = latent[b, t].unsqueeze(2)
    steps = t[:, None] + 1 + torch.arange(pred_length)[None, :]
    actual = history[b[:, None], steps].permute(0, 2, 1, 3)
    m = mask[b, t].to(history.dtype)
    mask_over = m[:, None, None, None].expand(-1, N, pred_length, d).contiguous()
    return input_traj, input_att, input_lat, actual, mask_over


def prediction_loss(gat_p, dec_p, input_traj, input_att, input_lat, actual, mask_over, noise,
                    drop_masks, drop_p, pred_length, use_behavior=True):
    """Loss of one agent in Prediction_policy.learn (nova/prediction_policy.py:203-225)."""
    S, N, _, d = input_traj.shape
    obs = input_traj.reshape(S, N, d)
    if use_behavior:
        obs = torch.cat([obs, input_lat.reshape(S, N, -1)], dim=-1)
    hid = gat_forward(gat_p, obs, input_att.reshape(S * N, -1), noise)
    pred = prediction_decoder_forward(dec_p, input_traj, hid, pred_length, drop_masks, drop_p)
    return masked_l1(actual, pred, mask_over, d * pred_length), pred


# ----------------------------------------------------------------------------------------------
# a9/a10 Behavior_policy.learn (soft update) (nova/stable_behavior_policy.py:128-157, 161-246)
# ----------------------------------------------------------------------------------------------
def behavior_windows(history, mask, j, L):
    """behavior_traj_wrapper: history [E,T,N,d], mask [E,T] ->
    (curr [E,N,L,d] right-aligned zero padded, next [E,N,L,d], mask_next [E,N,L,d])."""
    E, T, N, d = history.shape
    start = max(0, j - L + 1)
    plug = max(0, L - j - 1)
    curr = torch.zeros(E, N, L, d, dtype=history.dtype)
    curr[:, :, plug:] = history[:, start:j + 1].permute(0, 2, 1, 3)
    nxt = history[:, j + 1:j + L + 1].permute(0, 2, 1, 3)
    mn = mask[:, j + 1:j + L + 1].to(history.dtype)[:, None, :, None].expand(E, N, L, d)
    return curr, nxt, mn


def behavior_learn_loss(enc_p, dec_p, history, mask, L, coef, drop_masks, drop_p,
                        penalty=0.0, thres=0.005, env_slice=None):
    """One agent's loss in Behavior_policy.learn.  history [E,T,N,d] (already [:, :-1]),
    mask [E,T] with the env-dependent polarity already applied, drop_masks [J, E*N, L, Hd].
    Returns (behavior_error, stability_error, loss).

    ``env_slice`` (a slice of the env axis): the SHARE of those envs in the three values.  The loss is a sum over envs --
    every window's numerator sum|next - pred| m is, its normaliser sum(m) + EPS depends on the masks alone and the stability
    term is a plain sum / E / L -- and the envs never interact (carried states and latents are per (env, entity) row), so the
    shares of a partition of the envs add up to the whole-batch values and their gradients to the whole-batch gradient.  Used
    where the autograd graph of the whole batch does not fit the host (256 envs x 79 windows in fp64 ~ 120 GB); the CPU suite
    checks the partition against the whole (tests/test_oracle_golden.py)."""
    if env_slice is not None:
        return _behavior_learn_share(enc_p, dec_p, history, mask, L, coef, drop_masks, drop_p, penalty, thres, env_slice)
    E, T, N, d = history.shape
    Z = enc_p["out.weight"].shape[0]
    R = enc_p["rnn.weight_hh_l0"].shape[1]
    Hd = dec_p["decoder.rnn.weight_hh_l0"].shape[1]
    dp = strip_prefix(dec_p, "decoder.")
    J = T - 1 - L
    latent = torch.zeros(E, N, Z, dtype=history.dtype)
    eh = torch.zeros(E * N, R, dtype=history.dtype)
    dh = torch.zeros(E * N, Hd, dtype=history.dtype)
    beh = 0.0
    stab = 0.0
    for j in range(J):
        curr, nxt, mn = behavior_windows(history, mask, j, L)
        dec_in = torch.cat([curr, latent[:, :, None, :].expand(E, N, L, Z)], dim=-1)      # behavior_net.py:63-66
        pred, dh = decoder_forward(dp, dec_in.reshape(E * N, L, d + Z), dh,
                                   None if drop_masks is None else drop_masks[j], drop_p)
        pred = pred.reshape(E, N, L, d)
        _, eh, new_lat = encoder_forward(enc_p, curr.reshape(E * N, L, d), eh)
        st = torch.linalg.norm(curr - pred, dim=-1).reshape(-1)
        latent = (1.0 - coef) * latent + new_lat.reshape(E, N, Z) * coef                  # :230
        beh = beh + masked_l1(nxt, pred, mn, d * N)                                       # :233-235
        stab = stab + torch.clamp(st - thres, min=0).sum() / E / L                        # :238-240
    beh = beh / J
    stab = stab / J
    return beh, stab, beh + penalty * stab


def _behavior_learn_share(enc_p, dec_p, history, mask, L, coef, drop_masks, drop_p, penalty, thres, sl):
    """behavior_learn_loss restricted to the envs ``sl`` under the WHOLE batch's normalisers (see there)"""
    E_all, T, N, d = history.shape
    h, m = history[sl], mask[sl]
    E = h.shape[0]
    lo = range(E_all)[sl][0]
    Z = enc_p["out.weight"].shape[0]
    R = enc_p["rnn.weight_hh_l0"].shape[1]
    Hd = dec_p["decoder.rnn.weight_hh_l0"].shape[1]
    dp = strip_prefix(dec_p, "decoder.")
    J = T - 1 - L
    latent = torch.zeros(E, N, Z, dtype=h.dtype)
    eh = torch.zeros(E * N, R, dtype=h.dtype)
    dh = torch.zeros(E * N, Hd, dtype=h.dtype)
    beh = 0.0
    stab = 0.0
    for j in range(J):
        curr, nxt, mn = behavior_windows(h, m, j, L)
        den = mask[:, j + 1:j + L + 1].to(h.dtype).sum() * (N * d)                          # sum of the expanded mask, all envs
        dec_in = torch.cat([curr, latent[:, :, None, :].expand(E, N, L, Z)], dim=-1)
        dm = None if drop_masks is None else drop_masks[j][lo * N:(lo + E) * N].to(h.dtype)
        pred, dh = decoder_forward(dp, dec_in.reshape(E * N, L, d + Z), dh, dm, drop_p)
        pred = pred.reshape(E, N, L, d)
        _, eh, new_lat = encoder_forward(enc_p, curr.reshape(E * N, L, d), eh)
        st = torch.linalg.norm(curr - pred, dim=-1).reshape(-1)
        latent = (1.0 - coef) * latent + new_lat.reshape(E, N, Z) * coef
        beh = beh + (torch.abs(nxt - pred) * mn).sum() / (den + EPS) * (d * N)
        stab = stab + torch.clamp(st - thres, min=0).sum() / E_all / L
    beh = beh / J
    stab = stab / J
    return beh, stab, beh + penalty * stab


# ----------------------------------------------------------------------------------------------
# a7  Behavior_policy.latent_update (rollout) (nova/stable_behavior_policy.py:83-123)
# ----------------------------------------------------------------------------------------------
def latent_update(enc_ps, history, enc_hidden, prev_latent, coef):
    """history [E,nA,N,L,d], enc_hidden [E,1,nA,N,R], prev_latent [E,nA,N,Z] ->
    (new_latent [E,nA,N,Z], new_hidden [E,1,nA,N,R])."""
    E, nA, N, L, d = history.shape
    lat, hid = [], []
    for i in range(nA):
        h0 = enc_hidden[:, 0, i].reshape(E * N, -1)
        _, hL, z = encoder_forward(enc_ps[i], history[:, i].reshape(E * N, L, d), h0)
        lat.append(z.reshape(E, N, -1))
        hid.append(hL.reshape(E, N, -1))
    new_lat = (1.0 - coef) * prev_latent + torch.stack(lat, 1) * coef
    return new_lat, torch.stack(hid, 1).unsqueeze(1)


# ----------------------------------------------------------------------------------------------
# a14 actor / critic (modules/agents/ippo_actor.py, modules/critics/ippo_critic.py,
#     utils/mappo_utils/{mlp,rnn,act,distributions,popart}.py)
# ----------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


RELU_HINT_LOG = []      # one entry per hinted ReLU evaluation: (units within delta of the kink, of those: branch taken from the hint differs)


def _relu_hinted(z, hint, delta):
    """ReLU whose branch at units within ``delta * max|z|`` of the kink is taken from ``hint`` (bool, same shape): at z = 0 both
    one-sided derivatives are valid, and an fp32 forward pass whose pre-activation differs from this one by a rounding error
    may sit on the other side.  The parity tests hand in the branch the implementation under test took (its saved
    activation > 0) so that a gradient is compared on the SAME branch; everywhere else the hint is ignored."""
    if hint is None:
        return torch.relu(z)
    own = z.detach() > 0
    amb = z.detach().abs() <= delta * z.detach().abs().max()
    RELU_HINT_LOG.append((int(amb.sum()), int((amb & (hint != own)).sum())))
    return z * torch.where(amb, hint, own).to(z.dtype)


def ac_trunk(p, x, h, relu_hint=None, hint_delta=1e-5, use_relu=True):
    """MLPBase (mlp.py:44-52, 24-28; layer_N = 1) + RNNLayer (rnn.py:24-27,77).
    x [R,F], h [R,M] -> (features [R,M], h' [R,M]).  ``relu_hint`` = (fc1 branch, fc2 branch): see _relu_hinted.
    ``use_relu`` = args.use_ReLU: active_func = [nn.Tanh(), nn.ReLU()][use_ReLU] (mlp.py:10)."""
    h1, h2 = relu_hint if relu_hint is not None else (None, None)
    act = (lambda z, hint: _relu_hinted(z, hint, hint_delta)) if use_relu else (lambda z, hint: torch.tanh(z))
    f = layer_norm(x, p["base.feature_norm.weight"], p["base.feature_norm.bias"])
    f = act(f @ p["base.mlp.fc1.0.weight"].t() + p["base.mlp.fc1.0.bias"], h1)
    f = layer_norm(f, p["base.mlp.fc1.2.weight"], p["base.mlp.fc1.2.bias"])
    f = act(f @ p["base.mlp.fc2.0.0.weight"].t() + p["base.mlp.fc2.0.0.bias"], h2)
    f = layer_norm(f, p["base.mlp.fc2.0.2.weight"], p["base.mlp.fc2.0.2.bias"])
    hn = gru_cell(f, h, p["rnn.rnn.weight_ih_l0"], p["rnn.rnn.weight_hh_l0"],
                  p["rnn.rnn.bias_ih_l0"], p["rnn.rnn.bias_hh_l0"])
    return layer_norm(hn, p["rnn.norm.weight"], p["rnn.norm.bias"]), hn


def actor_logits(p, x, h, avail=None, relu_hint=None, use_relu=True):
    f, hn = ac_trunk(p, x, h, relu_hint, use_relu=use_relu)
    logits = f @ p["act.action_out.linear.weight"].t() + p["act.action_out.linear.bias"]
    if avail is not None:
        logits = torch.where(avail == 0, torch.full_like(logits, -1e10), logits)   # distributions.py:66-67
    return logits, hn


def actor_evaluate(p, x, h, actions, avail=None, relu_hint=None, use_relu=True):
    """R_Actor.evaluate_actions (ippo_actor.py:74-102, act.py:159-164):
    -> (logp [R,1], entropy scalar = unmasked mean over rows)."""
    logits, _ = actor_logits(p, x, h, avail, relu_hint, use_relu=use_relu)
    logp_all = torch.log_softmax(logits, dim=-1)
    logp = logp_all.gather(-1, actions.long().reshape(-1, 1))
    pr = logp_all.exp()
    ent = -(pr * torch.clamp(logp_all, min=torch.finfo(logp_all.dtype).min)).sum(-1).mean()
    return logp, ent


def critic_value(p, x, h, relu_hint=None, use_relu=True):
    """R_Critic.forward (ippo_critic.py:47-65); PopArt.forward is a plain Linear (popart.py:41-46)."""
    f, hn = ac_trunk(p, x, h, relu_hint, use_relu=use_relu)
    return f @ p["v_out.weight"].t() + p["v_out.bias"], hn


def build_inputs_rollout(history_t, att_t, beh_t, last_onehot_t, n_agents, gat=True, beh=True):
    """DcntrlMAC._build_inputs (controllers/dcntrl_controller.py:187-213).
    history_t [E,nA,N,d] ... last_onehot_t [E,nA,n_act] (zeros at t=0) -> [E,nA,F]."""
    E = history_t.shape[0]
    parts = [history_t]
    if gat:
        parts.append(att_t)
    if beh:
        parts.append(beh_t)
    states = torch.cat(parts, dim=-1).reshape(E, n_agents, -1)
    eye = torch.eye(n_agents, dtype=history_t.dtype)[None].expand(E, -1, -1)
    return torch.cat([states, last_onehot_t.to(history_t.dtype), eye], dim=-1)


def build_inputs_train(agent_id, history, att, beh_lat, actions_onehot, n_agents, gat=True, beh=True):
    """DcntrlMAC._build_inputs_ippo (dcntrl_controller.py:87-115): per-agent tensors
    history [bs,T1,N,d] ... actions_onehot [bs,T1,n_act] -> [bs,T1,F]."""
    bs, T1 = history.shape[:2]
    parts = [history]
    if gat:
        parts.append(att)
    if beh:
        parts.append(beh_lat)
    states = torch.cat(parts, dim=-1).reshape(bs, T1, -1)
    last = torch.cat([actions_onehot[:, :1], actions_onehot[:, :-1]], dim=1).to(history.dtype)
    idoh = torch.zeros(bs, T1, n_agents, dtype=history.dtype)
    idoh[:, :, agent_id] = 1
    return torch.cat([states, last, idoh], dim=-1)


# ----------------------------------------------------------------------------------------------
# a16-a18 PPO pieces (learners/ippo_learner.py, utils/mappo_utils/util.py)
# ----------------------------------------------------------------------------------------------
def gae_returns(rewards, values_all, masks_all, gamma, lam, use_gae=True):
    """compute_returns (ippo_learner.py:344-365).  rewards [bs,T,1], values_all [bs,T+1,1],
    masks_all [bs,T+1,1] (= 1 - terminated) -> returns [bs,T,1].  use_gae False (:360-362): discounted returns bootstrapped
    from the last value."""
    T = rewards.shape[1]
    if not use_gae:
        out, nxt = [None] * T, values_all[:, T]
        for t in reversed(range(T)):
            nxt = nxt * gamma * masks_all[:, t + 1] + rewards[:, t]
            out[t] = nxt
        return torch.stack(out, dim=1)
    gae = torch.zeros_like(rewards[:, 0])
    out = [None] * T
    for t in reversed(range(T)):
        delta = rewards[:, t] + gamma * values_all[:, t + 1] * masks_all[:, t + 1] - values_all[:, t]
        gae = delta + gamma * lam * masks_all[:, t + 1] * gae
        out[t] = gae + values_all[:, t]
    return torch.stack(out, dim=1)


def normalise_advantages(returns, values, masks):
    """ippo_learner.py:273-279: zero dead steps, unbiased std over ALL entries."""
    adv = (returns - values).clone()
    adv[masks == 0.0] = 0.0
    std, mean = torch.std_mean(adv)
    return (adv - mean) / (std + 1e-5)


def huber_loss(e, d):
    """utils/mappo_utils/util.py:33-36 -- one-sided on purpose (b = e > d, not |e| > d)."""
    a = (e.abs() <= d).to(e.dtype)
    b = (e > d).to(e.dtype)
    return a * e ** 2 / 2 + b * d * (e.abs() - d / 2)


def ppo_losses(logp, ent, values, old_logp, adv, value_preds, returns, masks,
               clip=0.2, huber_delta=10.0, ent_coef=0.01, vcoef=0.5,
               use_huber_loss=True, use_clipped_value_loss=True, use_value_active_masks=True, use_policy_active_masks=True):
    """ippo_learner.py:185-197 (policy) and :128-159 (value).  All [R,1].
    -> (actor_objective, policy_loss, critic_objective, value_loss, ratio)."""
    ratio = torch.exp(logp - old_logp)
    s1 = ratio * adv
    s2 = torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv
    surr = -torch.min(s1, s2).sum(-1, keepdim=True)
    pol = (surr * masks).sum() / masks.sum() if use_policy_active_masks else surr.mean()
    vclip = value_preds + (values - value_preds).clamp(-clip, clip)
    loss = (lambda e: huber_loss(e, huber_delta)) if use_huber_loss else (lambda e: e ** 2 / 2)      # envs/util.py:28-29
    vl = loss(returns - values)
    if use_clipped_value_loss:
        vl = torch.max(vl, loss(returns - vclip))
    vloss = (vl * masks).sum() / masks.sum() if use_value_active_masks else vl.mean()
    return pol - ent * ent_coef, pol, vloss * vcoef, vloss, ratio


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics: total L2 norm; scale by
    min(1, max_norm/(norm+1e-6)).  grads: list of tensors (modified in place)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).to(grads[0].dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adam_step(p, g, m, v, step, lr, eps, b1=0.9, b2=0.999):
    """torch.optim.Adam (weight_decay 0, amsgrad False) single-tensor update, in place.
    ``step`` is the 1-based step count AFTER increment."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# ----------------------------------------------------------------------------------------------
# iPLAN-Hard ablation: Behavior_policy.learn with hard latent updates (nova/behavior_policy.py:119-215)
# ----------------------------------------------------------------------------------------------
def behavior_hard_learn_loss(enc_p, dec_p, history, mask, L, drop_masks, drop_p):
    """history [E,T,N,d] (already [:, :-1]), mask [E,T] (polarity applied), drop_masks [J, E*N, L, Hd].
    Non-overlapping blocks of L steps; block j+1 is predicted from block j and the latent encoded from
    blocks <= j-1; the latent is replaced (not blended); ONE normaliser over all J*L masked steps, and the
    mask is taken at the CURRENT block's steps (:145-150, 181-187)."""
    E, T, N, d = history.shape
    Z = enc_p["out.weight"].shape[0]
    R = enc_p["rnn.weight_hh_l0"].shape[1]
    Hd = dec_p["decoder.rnn.weight_hh_l0"].shape[1]
    dp = strip_prefix(dec_p, "decoder.")
    nh = T // L
    J = nh - 1
    blocks = history[:, :nh * L].reshape(E, nh, L, N, d)
    latent = torch.zeros(E, N, Z, dtype=history.dtype)
    eh = torch.zeros(E * N, R, dtype=history.dtype)
    dh = torch.zeros(E * N, Hd, dtype=history.dtype)
    preds = []
    for j in range(J):
        curr = blocks[:, j].permute(0, 2, 1, 3)                                    # [E,N,L,d]
        dec_in = torch.cat([curr, latent[:, :, None, :].expand(E, N, L, Z)], dim=-1)
        pred, dh = decoder_forward(dp, dec_in.reshape(E * N, L, d + Z), dh, None if drop_masks is None else drop_masks[j], drop_p)
        preds.append(pred.reshape(E, N, L, d).permute(0, 2, 1, 3))               # [E,L,N,d]
        _, eh, latent = encoder_forward(enc_p, curr.reshape(E * N, L, d), eh)
        latent = latent.reshape(E, N, Z)
    nxt = blocks[:, 1:].reshape(E, J * L, N, d)
    pred = torch.stack(preds, 1).reshape(E, J * L, N, d)
    m = mask[:, :J * L].to(history.dtype)[:, :, None, None].expand(E, J * L, N, d)
    return (torch.abs(nxt - pred) * m).sum() / (m.sum() + EPS) * d * N


def mlp3(p, x, softmax=False):
    """nova/behavior_FC_net.py:15-19 / :32-36: out(tanh(linear_2(tanh(linear_1(x))))), optionally softmaxed."""
    h = torch.tanh(x @ p["linear_1.weight"].t() + p["linear_1.bias"])
    h = torch.tanh(h @ p["linear_2.weight"].t() + p["linear_2.bias"])
    y = h @ p["out.weight"].t() + p["out.bias"]
    return torch.softmax(y, dim=-1) if softmax else y


def behavior_fc_learn_loss(enc_p, dec_p, history, L):
    """nova/behavior_FC_policy.py:168-201 for one agent.  history [E,T,N,d] (already [:, :-1]).  Right-aligned zero-padded
    windows (behavior_traj_wrapper :110-138); the decoder of window j is fed the encoder output of window j-1 (zeros at
    j = 0); the "mask" over the next trajectory is all ones (the wrapper fills the CURRENT mask twice, :131-136), so every
    window's error is the unmasked sum / (E N L d + EPS) * d * N, averaged over the J = T-1-L windows."""
    E, T, N, d = history.shape
    Z = enc_p["out.weight"].shape[0]
    dp = strip_prefix(dec_p, "decoder.")
    J = T - 1 - L
    pad = torch.cat([torch.zeros(E, L - 1, N, d, dtype=history.dtype), history], dim=1)
    latent = torch.zeros(E, N, Z, dtype=history.dtype)
    total = 0.0
    for j in range(J):
        curr = pad[:, j:j + L].permute(0, 2, 1, 3).reshape(E, N, L * d)           # steps j-L+1 .. j
        nxt = pad[:, j + 1:j + 1 + L].permute(0, 2, 1, 3).reshape(E, N, L * d)    # steps j-L+2 .. j+1
        pred = mlp3(dp, torch.cat([curr, latent], dim=-1))
        latent = mlp3(enc_p, curr, softmax=True)
        total = total + torch.abs(nxt - pred).sum() / (float(E * N * L * d) + EPS) * d * N
    return total / J


# ----------------------------------------------------------------------------------------------
# a17 IPPOLearner.train for ONE agent (learners/ippo_learner.py:227-317): compute_returns -> advantage
#     normalisation -> ppo_epoch x (evaluate, losses, clip, Adam) on the first batch_size * T rows.
#     num_mini_batch == 1: every epoch is a full-batch step, so the randperm order does not matter.
# ----------------------------------------------------------------------------------------------
def ppo_train_agent(agent_id, actor_p, critic_p, fields, args, rows=None, row_index_lists=None, probe_last_step=None,
                    probe_relu_hint=None, probe_steps=None):
    """fields: dict of [E, T+1, nA, ...] episode tensors (the buffer content).  actor_p / critic_p: dicts of leaf
    tensors with requires_grad (updated IN PLACE by Adam).  ``row_index_lists``: optional list (per epoch) of lists of
    row-index tensors (minibatches, generate_data :368-424); default = one minibatch with the first ``rows`` rows.
    ``probe_last_step`` = (actor params, critic params) dicts: BEFORE the last optimiser step the clipped gradients of that
    step's minibatch are also evaluated AT these parameters (old log-probs / advantages / returns still those of the
    pre-train parameters, as in the reference) and returned as ``probe_grads`` -- the parity tests hand in the parameters
    the implementation under test actually held before its last step, so that a multi-step gradient is compared at ONE
    parameter point instead of across two trajectories that Adam's sign-like first steps have already separated.
    ``probe_relu_hint`` = ((actor fc1, fc2 branches), (critic fc1, fc2 branches)), bool [rows of the last minibatch, M]: the
    ReLU branches the implementation took in that step's forward pass, used by the probe evaluation at units within fp32
    rounding of the kink only (_relu_hinted).
    ``probe_steps`` = {step index k (0-based): ((actor params, critic params), relu hint or None)}: the same evaluation in front
    of ANY optimiser step of the run, results in ``probe_grads_by_step[k]`` (``hint_log_by_step[k]`` = the RELU_HINT_LOG entries
    of that evaluation) -- a gradient in the middle of the 15-epoch trajectory is checked at the implementation's own
    parameters exactly like the last one.
    Returns dict(returns, adv, old_logp, values_all, stats per epoch[, probe_grads, probe_grads_by_step])."""
    f, i = fields, agent_id
    gat, behv = args.GAT_enable, args.Behavior_enable
    E, T1 = f["history"].shape[:2]
    T = T1 - 1
    dt = f["history"].dtype
    M = f["rnn_states_actors"].shape[-1]
    if rows is None:
        rows = args.batch_size * T
    x_all = build_inputs_train(i, f["history"][:, :, i], f["attention_latent"][:, :, i] if gat else None,
                               f["behavior_latent"][:, :, i] if behv else None, f["actions_onehot"][:, :, i],
                               args.n_agents, gat, behv)
    masks_all = 1.0 - f["terminated"][:, :, i].to(dt)
    F_ = x_all.shape[-1]
    use_relu = bool(getattr(args, "use_ReLU", True))
    with torch.no_grad():
        v_all, _ = critic_value(critic_p, x_all.reshape(-1, F_), f["rnn_states_critics"][:, :, i].reshape(-1, M), use_relu=use_relu)
        v_all = v_all.reshape(E, T + 1, 1)
        rets = gae_returns(f["reward"][:, :-1, i].to(dt), v_all, masks_all, args.gamma, args.gae_lambda, getattr(args, "use_gae", True))
        adv = normalise_advantages(rets, v_all[:, :-1], masks_all[:, :-1])
        x = x_all[:, :-1].reshape(-1, F_)
        ha = f["rnn_states_actors"][:, :-1, i].reshape(-1, M)
        hc = f["rnn_states_critics"][:, :-1, i].reshape(-1, M)
        acts = f["actions"][:, :-1, i].reshape(-1, 1)
        avail = f["avail_actions"][:, :-1, i].reshape(-1, args.n_actions)
        old_logp, _ = actor_evaluate(actor_p, x, ha, acts, avail, use_relu=use_relu)
    ms = [{k: torch.zeros_like(v) for k, v in prm.items()} for prm in (actor_p, critic_p)]
    vs = [{k: torch.zeros_like(v) for k, v in prm.items()} for prm in (actor_p, critic_p)]
    steps = [0, 0]
    stats = []
    loss_kw = {k: getattr(args, k, True) for k in ("use_huber_loss", "use_clipped_value_loss", "use_value_active_masks", "use_policy_active_masks")}

    held = {}

    def objectives(ap, cp, sl, hints=(None, None)):
        logp, ent = actor_evaluate(ap, x[sl], ha[sl], acts[sl], avail[sl], relu_hint=hints[0], use_relu=use_relu)
        val, _ = critic_value(cp, x[sl], hc[sl], relu_hint=hints[1], use_relu=use_relu)
        if val.requires_grad:
            val.retain_grad()                                 # (probe_eval: the per-row d loss / d value, for the conditioning figure)
        held["val"] = val
        return ppo_losses(logp, ent, val, old_logp[sl], adv.reshape(-1, 1)[sl], v_all[:, :-1].reshape(-1, 1)[sl],
                          rets.reshape(-1, 1)[sl], masks_all[:, :-1].reshape(-1, 1)[sl],
                          args.clip_param, args.huber_delta, args.entropy_coef, args.value_loss_coef, **loss_kw) + (ent,)

    probe_grads = None
    n_steps = sum(len([slice(0, rows)] if row_index_lists is None else row_index_lists[ep]) for ep in range(args.ppo_epoch))
    probes = dict(probe_steps or {})
    if probe_last_step is not None:
        probes[n_steps - 1] = (probe_last_step, probe_relu_hint)
    by_step, hint_log_by_step = {}, {}
    cond_by_step = {}                                         # optimiser step -> conditioning figure of its probe (see probe_eval)

    def probe_eval(point, hint, sl):
        pa, pc = ({k: (v.detach().to(dt) if v.is_floating_point() else v.detach()).clone().requires_grad_(v.is_floating_point())
                   for k, v in src.items()} for src in point)
        a_obj, _, c_obj, _, _, _ = objectives(pa, pc, sl, hint if hint is not None else (None, None))
        a_obj.backward()
        c_obj.backward()
        # Conditioning of the gradients that are plain ROW SUMS of d loss / d value (critic v_out.bias; rnn.norm.bias = that sum times
        # v_out.weight): sum |g_row| / |sum g_row|.  When the residuals v - return nearly cancel over the rows this is >> 1 and
        # rounding of the values at the last bit, which no fp32 implementation avoids, moves those gradients by ~ cond x 2^-24 of
        # their own size (tests/oracle_checks.py uses it as a tolerance term)
        gv = held["val"].grad
        held["cond"] = float(gv.abs().sum() / gv.sum().abs().clamp_min(1e-300)) if gv is not None else 1.0
        out = []
        for prm in (pa, pc):
            trainable = [k for k in prm if prm[k].grad is not None]
            clip_grad_norm([prm[k].grad for k in trainable], args.max_grad_norm)
            out.append({k: prm[k].grad for k in trainable})
        return out

    for ep in range(args.ppo_epoch):
        batches = [slice(0, rows)] if row_index_lists is None else row_index_lists[ep]
        for sl in batches:
            if steps[0] in probes:
                n0 = len(RELU_HINT_LOG)
                by_step[steps[0]] = probe_eval(probes[steps[0]][0], probes[steps[0]][1], sl)
                cond_by_step[steps[0]] = held["cond"]
                hint_log_by_step[steps[0]] = list(RELU_HINT_LOG[n0:])
                if steps[0] == n_steps - 1 and probe_last_step is not None:
                    probe_grads = by_step[steps[0]]
            for prm in (actor_p, critic_p):
                for v in prm.values():
                    v.grad = None
            a_obj, pol, c_obj, vl, ratio, ent = objectives(actor_p, critic_p, sl)
            a_obj.backward()
            c_obj.backward()
            norms = []
            for gi, prm in enumerate((actor_p, critic_p)):
                trainable = [k for k in prm if prm[k].grad is not None]
                norms.append(float(clip_grad_norm([prm[k].grad for k in trainable], args.max_grad_norm)))
                steps[gi] += 1
                with torch.no_grad():
                    for k in trainable:
                        adam_step(prm[k], prm[k].grad, ms[gi][k], vs[gi][k], steps[gi],
                                  args.lr if gi == 0 else args.critic_lr, args.optim_eps)
            stats.append(dict(policy_loss=float(pol.detach()), value_loss=float(vl.detach()), entropy=float(ent.detach()), ratio=float(ratio.detach().mean()),
                              actor_grad_norm=norms[0], critic_grad_norm=norms[1]))
    return dict(returns=rets, adv=adv, old_logp=old_logp, values_all=v_all, stats=stats, probe_grads=probe_grads,
                probe_grads_by_step=by_step, hint_log_by_step=hint_log_by_step,
                value_grad_row_sum_cond=cond_by_step)


# ----------------------------------------------------------------------------------------------
# a19 Seq2Seq.forward (nova/Seq2Seq.py:52-70): stacked-GRU encoder from a zero state, autoregressive decoder
# ----------------------------------------------------------------------------------------------
def seq2seq_forward(p, in_data, last_location, pred_length, teacher=None, coins=None, drop_masks=None, drop_p=0.0):
    """p: Seq2Seq state_dict; in_data [R,T,C], last_location [R,1,O], teacher [R,P,O] / coins [P] bools (teacher forcing of the
    NEXT step's input) or None, drop_masks [P, R, 1, H] keep flags or None -> [R, P, O]."""
    layers = sum(1 for k in p if k.startswith("encoder.rnn.weight_ih_l"))
    H = p["encoder.rnn.weight_hh_l0"].shape[1]
    R = in_data.shape[0]
    h = [torch.zeros(R, H, dtype=in_data.dtype) for _ in range(layers)]
    for t in range(in_data.shape[1]):
        x = in_data[:, t]
        for k in range(layers):
            h[k] = gru_cell(x, h[k], p[f"encoder.rnn.weight_ih_l{k}"], p[f"encoder.rnn.weight_hh_l{k}"],
                            p[f"encoder.rnn.bias_ih_l{k}"], p[f"encoder.rnn.bias_hh_l{k}"])
            x = h[k]
    y = last_location[:, 0]
    outs = []
    for t in range(pred_length):
        x = y
        for k in range(layers):
            h[k] = gru_cell(x, h[k], p[f"decoder.rnn.weight_ih_l{k}"], p[f"decoder.rnn.weight_hh_l{k}"],
                            p[f"decoder.rnn.bias_ih_l{k}"], p[f"decoder.rnn.bias_hh_l{k}"])
            x = h[k]
        a = torch.tanh(x)
        if drop_masks is not None:
            a = a * drop_masks[t].reshape(R, H) / (1.0 - drop_p)
        out = a @ p["decoder.linear.weight"].t() + p["decoder.linear.bias"]
        outs.append(out)
        y = teacher[:, t] if (teacher is not None and coins is not None and coins[t]) else out
    return torch.stack(outs, dim=1)
