"""Generate tests/golden/*.pt by running the REFERENCE (imported from /root/reference) on seeded
synthetic inputs, and pin the oracle restatement (oracle/iplan_oracle.py) against it.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (the reference does not exist on the GPU
box); the fixtures it writes are committed so that ``-m gpu`` tests can check the HIP path
against real reference outputs without the reference being present.

    python oracle/make_golden.py            # regenerates every fixture, asserts oracle == reference

Randomness the reference draws internally (gumbel noise, dropout keep-masks, Categorical samples,
np.random.choice) is captured by patching torch.nn.functional.{gumbel_softmax,dropout} with
stream-identical recording versions, and stored in the fixture so every implementation consumes
the same values.
"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("IPLAN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import torch.nn.functional as F  # noqa: E402

from iplan_amd.config import default_args  # noqa: E402
from iplan_amd import synth  # noqa: E402
from oracle import iplan_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REC = {"gumbel": [], "dropout": []}
_orig_dropout = F.dropout
_orig_gumbel = F.gumbel_softmax


def _rec_gumbel(logits, tau=1, hard=False, eps=1e-10, dim=-1):
    g = -torch.empty_like(logits).exponential_().log()      # same draw as torch's implementation
    REC["gumbel"].append(g.detach().clone())
    return ((logits + g) / tau).softmax(dim)


def _rec_dropout(input, p=0.5, training=True, inplace=False):
    if not training or p == 0.0:
        return input
    keep = torch.empty_like(input).bernoulli_(1 - p)
    REC["dropout"].append(keep.detach().clone())
    return input * keep / (1 - p)


def patch():
    F.gumbel_softmax = _rec_gumbel
    F.dropout = _rec_dropout
    REC["gumbel"].clear()
    REC["dropout"].clear()


def unpatch():
    F.gumbel_softmax = _orig_gumbel
    F.dropout = _orig_dropout


def sd(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def to64(p):
    return {k: v.double() for k, v in p.items()}


def req(p):
    return {k: v.clone().requires_grad_(True) for k, v in p.items()}


def check(name, a, b, tol):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    err = (a.double() - b.double()).abs().max().item()
    scale = max(1.0, b.double().abs().max().item())
    status = "ok" if err <= tol * scale else "FAIL"
    print(f"  [{status}] {name}: max|d|={err:.3e} (scale {scale:.2e}, tol {tol:g})")
    assert err <= tol * scale, name


class NullLogger:
    def log_stat(self, *a, **k):
        pass


def small_args(**kw):
    a = default_args("highway", use_cuda=False, **kw)
    a.obs_shape = a.obs_shape_single * 5
    a.state_shape = a.obs_shape
    return a


# ----------------------------------------------------------------------------------------------
def golden_gat(tag, B, N, D, seed):
    from nova.GAT_Net import GAT_Net
    print(f"gat_{tag}: B={B} N={N} D={D}")
    args = small_args(max_vehicle_num=N)
    torch.manual_seed(seed)
    net = GAT_Net(D, args)
    gen = torch.Generator().manual_seed(seed + 1)
    obs = synth.make_history(gen, (B,), N, D)
    obs[..., 5:] = torch.rand(B, N, D - 5, generator=gen) if D > 5 else obs[..., 5:]
    h_prev = torch.randn(B * N, args.attention_dim, generator=gen) * 0.1
    gout = torch.randn(B * N, args.attention_dim, generator=gen)
    patch()
    torch.manual_seed(seed + 2)
    out = net(obs, h_prev)
    noise = REC["gumbel"][0]
    unpatch()
    (out * gout).sum().backward()
    grads = {k: v.grad.detach().clone() for k, v in net.named_parameters()}
    p = sd(net)
    # oracle fp32 and fp64 vs reference
    o32 = O.gat_forward(p, obs, h_prev, noise)
    check("oracle32 out", o32, out, 1e-5)
    pr = req(to64(p))
    o64 = O.gat_forward(pr, obs.double(), h_prev.double(), noise.double())
    check("oracle64 out", o64, out, 1e-5)
    (o64 * gout.double()).sum().backward()
    for k in grads:
        check("grad " + k, pr[k].grad, grads[k], 2e-4)
    # fp64 reference-quality grads are stored as the gradient golden (less fp32 noise than the
    # fp32 reference run; the fp32 run is stored too)
    torch.save(dict(B=B, N=N, D=D, params=p, obs=obs, h_prev=h_prev, noise=noise, gout=gout,
                    out=out.detach(), grads=grads,
                    out64=o64.detach(), grads64={k: v.grad.clone() for k, v in pr.items()}),
               os.path.join(GOLD, f"gat_{tag}.pt"))


def golden_encoder(seed=10):
    from nova.behavior_net import EncoderRNN
    print("encoder")
    R, L, d = 12, 10, 5
    torch.manual_seed(seed)
    net = EncoderRNN(d, 32, 8, 1)
    gen = torch.Generator().manual_seed(seed + 1)
    x = synth.make_history(gen, (R,), L, d)
    h0 = torch.randn(1, R, 32, generator=gen) * 0.1
    g_lat = torch.randn(R, 8, generator=gen)
    g_h = torch.randn(R, 32, generator=gen)
    out, hL, lat = net(x, h0)
    ((lat * g_lat).sum() + (hL[0] * g_h).sum()).backward()
    grads = {k: v.grad.detach().clone() for k, v in net.named_parameters()}
    p = sd(net)
    o_out, o_h, o_lat = O.encoder_forward(p, x, h0[0])
    check("seq", o_out, out, 1e-5)
    check("hL", o_h, hL[0], 1e-5)
    check("latent", o_lat, lat, 1e-5)
    pr = req(to64(p))
    _, h64, l64 = O.encoder_forward(pr, x.double(), h0[0].double())
    ((l64 * g_lat.double()).sum() + (h64 * g_h.double()).sum()).backward()
    for k in grads:
        check("grad " + k, pr[k].grad, grads[k], 1e-4)
    torch.save(dict(params=p, x=x, h0=h0[0], g_lat=g_lat, g_h=g_h, seq=out.detach(), hL=hL[0].detach(),
                    latent=lat.detach(), grads=grads, grads64={k: v.grad.clone() for k, v in pr.items()}),
               os.path.join(GOLD, "encoder.pt"))


def golden_decoder(seed=20):
    from nova.behavior_net import Behavior_Latent_Decoder
    print("behavior decoder")
    E, N, L, d, Z, Hd = 2, 5, 10, 5, 8, 64
    torch.manual_seed(seed)
    net = Behavior_Latent_Decoder(d + Z, Hd, 1, d, 0.1)
    gen = torch.Generator().manual_seed(seed + 1)
    curr = synth.make_history(gen, (E, N), L, d)
    lat = torch.softmax(torch.randn(E, N, Z, generator=gen), -1)
    h0 = torch.randn(1, E * N, Hd, generator=gen) * 0.1
    g_y = torch.randn(E * N, L, d, generator=gen)
    g_h = torch.randn(E * N, Hd, generator=gen)
    patch()
    torch.manual_seed(seed + 2)
    y, hT = net(curr, lat, h0)
    mask = REC["dropout"][0]
    unpatch()
    ((y * g_y).sum() + (hT[0] * g_h).sum()).backward()
    grads = {k: v.grad.detach().clone() for k, v in net.named_parameters()}
    p = sd(net)
    dec_in = torch.cat([curr, lat[:, :, None, :].expand(E, N, L, Z)], -1).reshape(E * N, L, d + Z)
    oy, oh = O.decoder_forward(O.strip_prefix(p, "decoder."), dec_in, h0[0], mask, 0.1)
    check("y", oy, y, 1e-5)
    check("hT", oh, hT[0], 1e-5)
    pr = req(to64(p))
    y64, h64 = O.decoder_forward(O.strip_prefix(pr, "decoder."), dec_in.double(), h0[0].double(), mask.double(), 0.1)
    ((y64 * g_y.double()).sum() + (h64 * g_h.double()).sum()).backward()
    for k in grads:
        check("grad " + k, pr[k].grad, grads[k], 1e-4)
    torch.save(dict(params=p, curr=curr, latent=lat, dec_in=dec_in, h0=h0[0], mask=mask, g_y=g_y, g_h=g_h,
                    y=y.detach(), hT=hT[0].detach(), grads=grads,
                    grads64={k: v.grad.clone() for k, v in pr.items()}),
               os.path.join(GOLD, "decoder.pt"))


def golden_pred_decoder(seed=30):
    from nova.prediction_net import Prediction_Decoder
    print("prediction decoder")
    B, N, d, A, P = 3, 7, 5, 32, 5
    torch.manual_seed(seed)
    net = Prediction_Decoder(d, A, 1, d, P, dropout=0.1, teacher_forcing_ratio=0)
    gen = torch.Generator().manual_seed(seed + 1)
    last = synth.make_history(gen, (B,), N, d).unsqueeze(2)
    teacher = synth.make_history(gen, (B, N), P, d)
    hid = torch.randn(B * N, A, generator=gen) * 0.3
    hid.requires_grad_(True)
    g_y = torch.randn(B, N, P, d, generator=gen)
    patch()
    torch.manual_seed(seed + 2)
    np.random.seed(seed)
    pred = net(last, teacher, hid)
    masks = torch.stack(REC["dropout"])           # [P, B*N, 1, A]
    unpatch()
    (pred * g_y).sum().backward()
    grads = {k: v.grad.detach().clone() for k, v in net.named_parameters()}
    g_hid = hid.grad.detach().clone()
    p = sd(net)
    o = O.prediction_decoder_forward(p, last, hid.detach(), P, masks, 0.1)
    check("pred", o, pred, 1e-5)
    pr = req(to64(p))
    h64 = hid.detach().double().requires_grad_(True)
    o64 = O.prediction_decoder_forward(pr, last.double(), h64, P, masks.double(), 0.1)
    (o64 * g_y.double()).sum().backward()
    for k in grads:
        check("grad " + k, pr[k].grad, grads[k], 1e-4)
    check("grad hidden", h64.grad, g_hid, 1e-4)
    torch.save(dict(params=p, last=last, hidden=hid.detach(), masks=masks, g_y=g_y, pred=pred.detach(),
                    grads=grads, g_hidden=g_hid, grads64={k: v.grad.clone() for k, v in pr.items()},
                    g_hidden64=h64.grad.clone()),
               os.path.join(GOLD, "pred_decoder.pt"))


def ref_episode_batch(args, E, seed, terminated_p):
    """Build the reference's own EpisodeBatch and fill it with the synthetic fields."""
    from components.episode_buffer import EpisodeBatch
    from components.transforms import OneHot
    scheme = synth.make_scheme(args)
    scheme.pop("actions_onehot")
    scheme.pop("filled")
    groups = {"agents": args.n_agents}
    preprocess = {"actions": ("actions_onehot", [OneHot(out_dim=args.n_actions)])}
    batch = EpisodeBatch(scheme, groups, E, args.episode_limit + 1, preprocess=preprocess, device="cpu")
    fields = synth.make_episode_fields(args, E, seed, terminated_p)
    for k, v in fields.items():
        batch.data.transition_data[k].copy_(v.view_as(batch.data.transition_data[k]))
    return batch, fields


def golden_prediction_learn(seed=40):
    from nova.prediction_policy import Prediction_policy
    print("prediction learn")
    args = small_args(max_vehicle_num=7, n_agents=2, episode_limit=20, pred_batch_size=8, batch_size_run=4)
    E = 4
    torch.manual_seed(seed)
    pol = Prediction_policy(args, NullLogger())
    batch, fields = ref_episode_batch(args, E, seed + 1, 0.9)
    pre = dict(gat=[sd(m) for m in pol.pred_GAT], dec=[sd(m) for m in pol.pred_decoder])
    patch()
    torch.manual_seed(seed + 2)
    np.random.seed(seed + 3)
    losses = pol.learn(batch, 0)
    gumbel = [g.clone() for g in REC["gumbel"]]
    drops = [d.clone() for d in REC["dropout"]]
    unpatch()
    post = dict(gat=[sd(m) for m in pol.pred_GAT], dec=[sd(m) for m in pol.pred_decoder])
    clipped = dict(gat=[{k: v.grad.clone() for k, v in m.named_parameters()} for m in pol.pred_GAT],
                   dec=[{k: v.grad.clone() for k, v in m.named_parameters()} for m in pol.pred_decoder])
    # replay with the oracle
    np.random.seed(seed + 3)
    hist = fields["history"][:, :-1]
    att = fields["attention_latent"][:, :-1]
    lat = fields["behavior_latent"][:, :-1]
    term = fields["terminated"][:, :-1]
    T = hist.shape[1]
    P = args.pred_length
    sel_all = []
    for i in range(args.n_agents):
        sel = np.random.choice(E * (T - P - 1), size=args.pred_batch_size, replace=False)
        for _ in range(P):
            np.random.random()
        sel_all.append(sel)
        it, ia, il, act, mo = O.prediction_gather(hist[:, :, i], att[:, :, i], lat[:, :, i],
                                                  term[:, :, i, 0], sel, P)
        gp = req(pre["gat"][i])
        dp = req(pre["dec"][i])
        masks = torch.stack(drops[i * P:(i + 1) * P])
        loss, _ = O.prediction_loss(gp, dp, it, ia, il, act, mo, gumbel[i], masks, args.decoder_dropout, P)
        check(f"agent{i} loss", loss, losses[i], 1e-5)
        loss.backward()
        gg = [gp[k].grad for k in gp]
        dg = [dp[k].grad for k in dp]
        O.clip_grad_norm(gg, args.max_grad_norm)
        O.clip_grad_norm(dg, args.max_grad_norm)
        for k in gp:
            check(f"agent{i} clipped grad gat.{k}", gp[k].grad, clipped["gat"][i][k], 2e-4)
        for k in dp:
            check(f"agent{i} clipped grad dec.{k}", dp[k].grad, clipped["dec"][i][k], 2e-4)
        for grp, prm in (("gat", gp), ("dec", dp)):
            for k in prm:
                w = prm[k].detach().clone()
                O.adam_step(w, prm[k].grad, torch.zeros_like(w), torch.zeros_like(w), 1,
                            args.lr_predict, args.optim_eps)
                check(f"agent{i} post {grp}.{k}", w, post[grp][i][k], 1e-6)
    torch.save(dict(args=vars(args), fields=fields, pre=pre, post=post, clipped=clipped,
                    losses=[float(x) for x in losses], gumbel=gumbel, dropout=drops,
                    select_idx=[torch.as_tensor(s) for s in sel_all], np_seed=seed + 3),
               os.path.join(GOLD, "prediction_learn.pt"))


def golden_behavior_learn(seed=50):
    from nova.stable_behavior_policy import Behavior_policy
    print("behavior learn")
    args = small_args(max_vehicle_num=5, n_agents=2, episode_limit=16, batch_size_run=3)
    E = 3
    torch.manual_seed(seed)
    pol = Behavior_policy(args, NullLogger())
    batch, fields = ref_episode_batch(args, E, seed + 1, 0.8)
    pre = dict(enc=[sd(m) for m in pol.behavior_encoder], dec=[sd(m) for m in pol.behavior_decoder])
    patch()
    torch.manual_seed(seed + 2)
    bl, sl, tl = pol.learn(batch, 0)
    drops = [d.clone() for d in REC["dropout"]]
    unpatch()
    post = dict(enc=[sd(m) for m in pol.behavior_encoder], dec=[sd(m) for m in pol.behavior_decoder])
    clipped = dict(enc=[{k: v.grad.clone() for k, v in m.named_parameters()} for m in pol.behavior_encoder],
                   dec=[{k: v.grad.clone() for k, v in m.named_parameters()} for m in pol.behavior_decoder])
    hist = fields["history"][:, :-1]
    term = fields["terminated"][:, :-1]
    T = hist.shape[1]
    L = args.max_history_len
    J = T - 1 - L
    for i in range(args.n_agents):
        ep = req(pre["enc"][i])
        dp = req(pre["dec"][i])
        masks = torch.stack(drops[i * J:(i + 1) * J])
        mask = term[:, :, i, 0]       # Highway polarity (stable_behavior_policy.py:190-193)
        beh, stab, loss = O.behavior_learn_loss(ep, dp, hist[:, :, i], mask, L, args.soft_update_coef,
                                                masks, args.decoder_dropout,
                                                args.behavior_variation_penalty, args.thres_small_variation)
        check(f"agent{i} behavior loss", beh, bl[i], 1e-5)
        check(f"agent{i} stability loss", stab, sl[i], 1e-5)
        loss.backward()
        O.clip_grad_norm([ep[k].grad for k in ep], args.max_grad_norm)
        O.clip_grad_norm([dp[k].grad for k in dp], args.max_grad_norm)
        for k in ep:
            check(f"agent{i} clipped grad enc.{k}", ep[k].grad, clipped["enc"][i][k], 2e-4)
        for k in dp:
            check(f"agent{i} clipped grad dec.{k}", dp[k].grad, clipped["dec"][i][k], 2e-4)
        for grp, prm in (("enc", ep), ("dec", dp)):
            for k in prm:
                w = prm[k].detach().clone()
                O.adam_step(w, prm[k].grad, torch.zeros_like(w), torch.zeros_like(w), 1,
                            args.lr_behavior, args.optim_eps)
                check(f"agent{i} post {grp}.{k}", w, post[grp][i][k], 1e-6)
    torch.save(dict(args=vars(args), fields=fields, pre=pre, post=post, clipped=clipped,
                    behavior_loss=[float(x) for x in bl], stability_loss=[float(x) for x in sl],
                    total_loss=[float(x) for x in tl], dropout=drops),
               os.path.join(GOLD, "behavior_learn.pt"))


def golden_rollout_step(seed=60):
    """GAT_latent_update + latent_update + select_actions_ippo on one vector step."""
    from nova.prediction_policy import Prediction_policy
    from nova.stable_behavior_policy import Behavior_policy
    from controllers.dcntrl_controller import DcntrlMAC
    print("rollout step")
    args = small_args(max_vehicle_num=7, n_agents=3, episode_limit=6, batch_size_run=4)
    E = 4
    torch.manual_seed(seed)
    pred = Prediction_policy(args, NullLogger())
    beh = Behavior_policy(args, NullLogger())
    scheme = synth.make_scheme(args)
    mac = DcntrlMAC(scheme, {"agents": args.n_agents}, args)
    batch, fields = ref_episode_batch(args, E, seed + 1, 0.9)
    hist_single, window = synth.rollout_step_inputs(args, E, seed + 2)
    gen = torch.Generator().manual_seed(seed + 3)
    att0 = (torch.randn(E, args.n_agents, args.max_vehicle_num, args.attention_dim, generator=gen) * 0.1).numpy()
    lat0 = torch.softmax(torch.randn(E, args.n_agents, args.max_vehicle_num, args.latent_dim, generator=gen), -1).numpy()
    eh0 = (torch.randn(E, 1, args.n_agents, args.max_vehicle_num, args.encoder_rnn_dim, generator=gen) * 0.1).numpy()
    patch()
    torch.manual_seed(seed + 4)
    att1 = pred.GAT_latent_update(hist_single, att0, lat0)
    gumbel = [g.clone() for g in REC["gumbel"]]
    unpatch()
    lat1, eh1 = beh.latent_update(window, eh0, lat0)
    # select actions: deterministic and sampled
    t_ep = 2
    vals_d, acts_d, logp_d, ha_d, hc_d = mac.select_actions_ippo(batch, t_ep, test_mode=True)
    torch.manual_seed(seed + 5)
    vals_s, acts_s, logp_s, ha_s, hc_s = mac.select_actions_ippo(batch, t_ep, test_mode=False)
    torch.manual_seed(seed + 5)
    q = torch.stack([torch.empty(E, args.n_actions).exponential_() for _ in range(args.n_agents)])
    vals_0, acts_0, logp_0, _, _ = mac.select_actions_ippo(batch, 0, test_mode=True)

    # oracle replay
    gat_p = [sd(m) for m in pred.pred_GAT]
    enc_p = [sd(m) for m in beh.behavior_encoder]
    act_p = [sd(m) for m in mac.agents]
    cri_p = [sd(m) for m in mac.critics]
    N = args.max_vehicle_num
    for i in range(args.n_agents):
        obs = torch.cat([torch.Tensor(hist_single[:, i]), torch.Tensor(lat0[:, i])], -1)
        o = O.gat_forward(gat_p[i], obs, torch.Tensor(att0[:, i]).reshape(E * N, -1), gumbel[i])
        check(f"GAT_latent_update agent{i}", o.reshape(E, N, -1), att1[:, i], 1e-5)
    ol, oh = O.latent_update(enc_p, torch.Tensor(window), torch.Tensor(eh0), torch.Tensor(lat0),
                             args.soft_update_coef)
    check("latent_update latent", ol, lat1, 1e-5)
    check("latent_update hidden", oh, eh1.detach(), 1e-5)
    f = fields
    for t, (vals, acts, logps, sampled) in ((t_ep, (vals_d, acts_d, logp_d, False)),
                                            (t_ep, (vals_s, acts_s, logp_s, True)),
                                            (0, (vals_0, acts_0, logp_0, False))):
        last = f["actions_onehot"][:, t - 1] if t > 0 else torch.zeros_like(f["actions_onehot"][:, 0])
        x = O.build_inputs_rollout(f["history"][:, t], f["attention_latent"][:, t], f["behavior_latent"][:, t],
                                   last, args.n_agents)
        for i in range(args.n_agents):
            logits, hn = O.actor_logits(act_p[i], x[:, i], f["rnn_states_actors"][:, t, i], f["avail_actions"][:, t, i])
            pr = torch.softmax(logits, -1)
            a = (pr / q[i]).argmax(-1) if sampled else pr.argmax(-1)
            assert torch.equal(a, torch.as_tensor(acts[:, i])), (t, sampled, i)
            lp = torch.log_softmax(logits, -1).gather(-1, a[:, None])
            check(f"t{t} sampled={sampled} logp agent{i}", lp, logps[i], 1e-5)
            v, hc = O.critic_value(cri_p[i], x[:, i], f["rnn_states_critics"][:, t, i])
            check(f"t{t} value agent{i}", v[:, 0], vals[:, i], 1e-5)
            if t == t_ep and not sampled:
                check(f"actor h agent{i}", hn, ha_d[0, :, i], 1e-5)
                check(f"critic h agent{i}", hc, hc_d[0, :, i], 1e-5)
    torch.save(dict(args=vars(args), fields=fields, hist_single=torch.as_tensor(hist_single),
                    window=torch.as_tensor(window), att0=torch.as_tensor(att0), lat0=torch.as_tensor(lat0),
                    eh0=torch.as_tensor(eh0), gumbel=gumbel, q=q, t_ep=t_ep,
                    gat=gat_p, enc=enc_p, actors=act_p, critics=cri_p,
                    att1=torch.as_tensor(att1), lat1=torch.as_tensor(lat1), eh1=eh1.detach(),
                    det=dict(values=torch.as_tensor(vals_d), actions=torch.as_tensor(acts_d),
                             logp=[x.detach() for x in logp_d], h_actor=torch.as_tensor(ha_d),
                             h_critic=torch.as_tensor(hc_d)),
                    smp=dict(values=torch.as_tensor(vals_s), actions=torch.as_tensor(acts_s),
                             logp=[x.detach() for x in logp_s]),
                    t0=dict(values=torch.as_tensor(vals_0), actions=torch.as_tensor(acts_0),
                            logp=[x.detach() for x in logp_0])),
               os.path.join(GOLD, "rollout_step.pt"))


def golden_ippo_train(seed=70, env="highway"):
    from controllers.dcntrl_controller import DcntrlMAC
    from learners.ippo_learner import IPPOLearner
    print(f"ippo train ({env})")
    if env == "highway":
        args = small_args(max_vehicle_num=7, n_agents=2, episode_limit=6, batch_size_run=4,
                          buffer_size=4, batch_size=3, ppo_epoch=3)
        tag = "ippo_train"
    elif env == "highway_tanh":
        # the second MLPBase activation (utils/mappo_utils/mlp.py:10, args.use_ReLU off): tanh in fc1 / fc2, tanh-gain init
        args = small_args(max_vehicle_num=6, n_agents=2, episode_limit=6, batch_size_run=4,
                          buffer_size=4, batch_size=3, ppo_epoch=3, use_ReLU=False)
        tag = "ippo_train_tanh"
    else:
        args = default_args("mpe_easy", use_cuda=False, episode_limit=8, batch_size_run=4,
                            buffer_size=4, batch_size=3, ppo_epoch=2)
        tag = "ippo_train_mpe"
    E = 4
    torch.manual_seed(seed)
    scheme = synth.make_scheme(args)
    mac = DcntrlMAC(scheme, {"agents": args.n_agents}, args)
    learner = IPPOLearner(mac, scheme, NullLogger(), args)
    batch, fields = ref_episode_batch(args, E, seed + 1, 0.15)
    pre = dict(actors=[sd(m) for m in mac.agents], critics=[sd(m) for m in mac.critics])
    learner.insert_episode_batch(batch)
    stats = {}

    class RecLogger:
        def log_stat(self, k, v, t):
            stats[k] = float(v)
    learner.logger = RecLogger()
    torch.manual_seed(seed + 2)
    learner.train(0)
    post = dict(actors=[sd(m) for m in mac.agents], critics=[sd(m) for m in mac.critics])

    # oracle replay (order-independent: one minibatch with all rows)
    for i in range(args.n_agents):
        ap = req(pre["actors"][i])
        cp = req(pre["critics"][i])
        O.ppo_train_agent(i, ap, cp, fields, args)
        for k in ap:
            check(f"agent{i} post actor.{k}", ap[k], post["actors"][i][k], 2e-5)
        for k in cp:
            check(f"agent{i} post critic.{k}", cp[k], post["critics"][i][k], 2e-5)
    torch.save(dict(args=vars(args), fields=fields, pre=pre, post=post, stats=stats),
               os.path.join(GOLD, f"{tag}.pt"))



from iplan_amd.synth import obs_stream  # noqa: E402


def golden_obs_wrapper(seed=80):
    """observation_wrapper.py: the reference class on a synthetic stream; outputs after every step + the episode output."""
    from types import SimpleNamespace
    import observation_wrapper as ref_ow                       # the reference's (REF is on sys.path)
    from oracle.obs_wrapper_oracle import HistoryWrapperOracle
    K, nA, obs_num, d, T, L, N = 3, 2, 6, 4, 12, 3, 16
    steps = obs_stream(K, nA, obs_num, d, T, seed)
    ref = ref_ow.observersation_state_history_wrapper(SimpleNamespace(obs_shape_single=d, batch_size_run=K), nA, N, T, L)
    orc = HistoryWrapperOracle(K, nA, N, T, L, d)
    ref.agent_obs_profile_init(steps[0])
    orc.init(steps[0])
    hist, single = [], []
    for o in steps:
        ref.obs_history_create(o)
        orc.create(o)
        hist.append(ref.obs_history_output().copy())
        single.append(ref.obs_single_history_output().copy())
        assert np.array_equal(hist[-1], orc.window(L)) and np.array_equal(single[-1], orc.single())
    mask = (np.random.default_rng(seed + 1).random((K, T, nA)) < 0.8).astype(np.float64)
    raw, seg = ref.obs_history_episode_output(mask)
    assert np.array_equal(raw, orc.window(T, mask))
    state = np.random.default_rng(seed + 2).uniform(-1, 1, (K, 1, 7 * (d + 1)))
    ns, no = ref.pure_obs_state_wrapper(state, steps[-1])
    torch.save(dict(dims=dict(K=K, nA=nA, obs_num=obs_num, d=d, T=T, L=L, N=N), steps=np.stack(steps), hist=np.stack(hist),
                    single=np.stack(single), mask=mask, raw=raw, seg=seg, vehicle_ids=ref.obs_vehicle_id, agent_ids=ref.agent_id,
                    state=state, new_state=ns, new_obs=no), os.path.join(GOLD, "obs_wrapper.pt"))


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(4)
    golden_gat("small", B=3, N=7, D=13, seed=1)
    golden_gat("hwy", B=2, N=55, D=13, seed=2)
    golden_gat("wide", B=2, N=20, D=40, seed=3)
    golden_encoder()
    golden_decoder()
    golden_pred_decoder()
    golden_prediction_learn()
    golden_behavior_learn()
    golden_rollout_step()
    golden_ippo_train(70, "highway")
    golden_ippo_train(80, "mpe_easy")
    golden_ippo_train(75, "highway_tanh")
    golden_checkpoint()
    golden_behavior_hard_learn()
    golden_obs_wrapper()
    golden_behavior_fc_learn()
    golden_wrappers()
    golden_seq2seq()
    print("all golden fixtures written to", GOLD)



def golden_checkpoint():
    """Known-answer test on the reference's OWN shipped checkpoints: trained IPPO actor /
    critic + Adam state of baselines/QMIX/marl_results/MPE/models/ippo_highway__seed=112358_03-12-23-44-17/40
    loaded into the reference classes (input dim 85), one forward on seeded inputs."""
    from modules.agents.ippo_actor import R_Actor
    from modules.critics.ippo_critic import R_Critic
    print("checkpoint fixture")
    root = os.path.join(REF, "baselines", "QMIX", "marl_results", "MPE", "models",
                        "ippo_highway__seed=112358_03-12-23-44-17", "40")
    args = default_args("highway", use_cuda=False)
    actor, critic = R_Actor(85, args), R_Critic(85, args)
    a_sd = torch.load(os.path.join(root, "agent_0.th"), map_location="cpu")
    c_sd = torch.load(os.path.join(root, "critic_0.th"), map_location="cpu")
    a_opt = torch.load(os.path.join(root, "actor_0_opt.th"), map_location="cpu")
    print(actor.load_state_dict(a_sd), critic.load_state_dict(c_sd))
    opt = torch.optim.Adam(actor.parameters(), lr=args.lr, eps=args.optim_eps)
    opt.load_state_dict(a_opt)
    gen = torch.Generator().manual_seed(90)
    B = 9
    x = torch.randn(B, 1, 85, generator=gen)
    h = torch.randn(1, B, 64, generator=gen) * 0.1
    avail = torch.ones(B, 1, 5)
    avail[::2, 0, 3] = 0
    with torch.no_grad():
        act, logp, h_a = actor(x, h, avail, deterministic=True)
        val, h_c = critic(x, h)
    torch.save(dict(actor=a_sd, critic=c_sd, actor_opt=a_opt, x=x, h=h, avail=avail, actions=act, logp=logp, h_actor=h_a,
                    values=val, h_critic=h_c), os.path.join(GOLD, "checkpoint_fixture.pt"))


def golden_behavior_hard_learn(seed=55):
    """iPLAN-Hard ablation (nova/behavior_policy.py)."""
    from nova.behavior_policy import Behavior_policy
    print("behavior learn (hard update)")
    args = small_args(max_vehicle_num=5, n_agents=2, episode_limit=25, batch_size_run=3, max_history_len=5)
    E = 3
    torch.manual_seed(seed)
    pol = Behavior_policy(args, NullLogger())
    batch, fields = ref_episode_batch(args, E, seed + 1, 0.8)
    pre = dict(enc=[sd(m) for m in pol.behavior_encoder], dec=[sd(m) for m in pol.behavior_decoder])
    patch()
    torch.manual_seed(seed + 2)
    bl = pol.learn(batch, 0)
    drops = [d.clone() for d in REC["dropout"]]
    unpatch()
    post = dict(enc=[sd(m) for m in pol.behavior_encoder], dec=[sd(m) for m in pol.behavior_decoder])
    clipped = dict(enc=[{k: v.grad.clone() for k, v in m.named_parameters()} for m in pol.behavior_encoder],
                   dec=[{k: v.grad.clone() for k, v in m.named_parameters()} for m in pol.behavior_decoder])
    hist = fields["history"][:, :-1]
    term = fields["terminated"][:, :-1]
    L = args.max_history_len
    J = hist.shape[1] // L - 1
    for i in range(args.n_agents):
        ep, dp = req(pre["enc"][i]), req(pre["dec"][i])
        masks = torch.stack(drops[i * J:(i + 1) * J])
        loss = O.behavior_hard_learn_loss(ep, dp, hist[:, :, i], term[:, :, i, 0], L, masks, args.decoder_dropout)
        check(f"agent{i} hard behavior loss", loss, bl[i], 1e-5)
        loss.backward()
        O.clip_grad_norm([ep[k].grad for k in ep], args.max_grad_norm)
        O.clip_grad_norm([dp[k].grad for k in dp], args.max_grad_norm)
        for k in ep:
            check(f"agent{i} clipped grad enc.{k}", ep[k].grad, clipped["enc"][i][k], 2e-4)
        for k in dp:
            check(f"agent{i} clipped grad dec.{k}", dp[k].grad, clipped["dec"][i][k], 2e-4)
    torch.save(dict(args=vars(args), fields=fields, pre=pre, post=post, clipped=clipped,
                    behavior_loss=[float(x) for x in bl], dropout=drops),
               os.path.join(GOLD, "behavior_hard_learn.pt"))


def golden_behavior_fc_learn(seed=58):
    """iPLAN-FC ablation (nova/behavior_FC_policy.py): three-layer perceptrons, no recurrent state."""
    from nova.behavior_FC_policy import Behavior_policy
    print("behavior learn (FC ablation)")
    args = small_args(max_vehicle_num=5, n_agents=2, episode_limit=14, batch_size_run=3, max_history_len=4)
    E = 3
    torch.manual_seed(seed)
    pol = Behavior_policy(args, NullLogger())
    batch, fields = ref_episode_batch(args, E, seed + 1, 0.8)
    pre = dict(enc=[sd(m) for m in pol.behavior_encoder], dec=[sd(m) for m in pol.behavior_decoder])
    win = torch.rand(E, args.n_agents, args.max_vehicle_num, args.max_history_len, args.obs_shape_single) * 2 - 1
    lat_roll, _ = pol.latent_update(win.numpy(), None, None)
    bl, _, _ = pol.learn(batch, 0)
    post = dict(enc=[sd(m) for m in pol.behavior_encoder], dec=[sd(m) for m in pol.behavior_decoder])
    clipped = dict(enc=[{k: v.grad.clone() for k, v in m.named_parameters()} for m in pol.behavior_encoder],
                   dec=[{k: v.grad.clone() for k, v in m.named_parameters()} for m in pol.behavior_decoder])
    hist = fields["history"][:, :-1]
    for i in range(args.n_agents):
        ep, dp = req(pre["enc"][i]), req(pre["dec"][i])
        loss = O.behavior_fc_learn_loss(ep, dp, hist[:, :, i], args.max_history_len)
        check(f"agent{i} FC behavior loss", loss, bl[i], 1e-5)
        loss.backward()
        O.clip_grad_norm([ep[k].grad for k in ep], args.max_grad_norm)
        O.clip_grad_norm([dp[k].grad for k in dp], args.max_grad_norm)
        for k in ep:
            check(f"agent{i} clipped grad enc.{k}", ep[k].grad, clipped["enc"][i][k], 2e-4)
        for k in dp:
            check(f"agent{i} clipped grad dec.{k}", dp[k].grad, clipped["dec"][i][k], 2e-4)
        check(f"agent{i} FC rollout latent", O.mlp3(pre["enc"][i], win[:, i].reshape(E, args.max_vehicle_num, -1), softmax=True),
              torch.as_tensor(lat_roll[:, i]), 1e-5)
    torch.save(dict(args=vars(args), fields=fields, pre=pre, post=post, clipped=clipped, behavior_loss=[float(x) for x in bl],
                    window=win, latent=torch.as_tensor(lat_roll)), os.path.join(GOLD, "behavior_fc_learn.pt"))


def golden_wrappers(seed=90):
    """behavior_traj_wrapper (stable_behavior_policy.py:128-157) and prediction_batch_wrapper (prediction_policy.py:122-164)."""
    from nova.stable_behavior_policy import Behavior_policy
    from nova.prediction_policy import Prediction_policy
    args = small_args(max_vehicle_num=4, n_agents=2, episode_limit=16, batch_size_run=3, max_history_len=4, pred_batch_size=6)
    torch.manual_seed(seed)
    E, T, N, d = 3, args.episode_limit, args.max_vehicle_num, args.obs_shape_single
    history = torch.rand(E, T, N, d) * 2 - 1
    attention = torch.randn(E, T, N, args.attention_dim) * 0.1
    latent = torch.softmax(torch.randn(E, T, N, args.latent_dim), -1)
    mask = (torch.rand(E, T) < 0.7).float()
    beh = Behavior_policy(args, NullLogger())
    traj = {step: tuple(t.clone() for t in beh.behavior_traj_wrapper(history, step, mask)) for step in (0, 2, 3, 7, T - 2 - args.max_history_len)}
    pred = Prediction_policy(args, NullLogger())
    np.random.seed(seed)
    out = pred.prediction_batch_wrapper(history, attention, mask, latent)
    torch.save(dict(args=vars(args), history=history, attention=attention, latent=latent, mask=mask, traj=traj, np_seed=seed,
                    pred=tuple(t.clone() for t in out)), os.path.join(GOLD, "wrappers.pt"))


def golden_seq2seq(seed=95):
    """nova/Seq2Seq.py: the reference class (2 layers x 64, teacher forcing 0.5, dropout active) and a 1 x 32 variant."""
    from nova.Seq2Seq import Seq2Seq
    print("seq2seq")
    cases = []
    for tag, (C, H, layers, P, No, ratio) in (("l2h64", (4, 64, 2, 6, 2, 0.5)), ("l1h32", (7, 32, 1, 4, 3, 0.0))):
        torch.manual_seed(seed)
        net = Seq2Seq(C, H, layers, P, num_node=5, output_size=No, dropout=0.5, teacher_forcing_ratio=ratio)
        gen = torch.Generator().manual_seed(seed + 1)
        R, T = 35, 6
        x = torch.rand(R, T, C, generator=gen) * 2 - 1
        last = torch.rand(R, 1, No, generator=gen) * 2 - 1
        teacher = torch.rand(R, P, No, generator=gen) * 2 - 1
        patch()
        torch.manual_seed(seed + 2)
        np.random.seed(seed + 3)
        # the reference's own autograd under a fixed linear loss sum(out * gw): the gradients a training loop over this module would get
        gw = torch.rand(R, P, No, generator=gen) * 2 - 1
        out = net(x, last, teacher)
        (out * gw).sum().backward()
        grads = {k: v.grad.clone() for k, v in net.named_parameters()}
        out = out.detach()
        masks = torch.stack(REC["dropout"])                                # [P, R, 1, H]
        unpatch()
        np.random.seed(seed + 3)
        coins = [bool(np.random.random() < ratio) for _ in range(P)]
        p = sd(net)
        o = O.seq2seq_forward(p, x, last, P, teacher, coins, masks, 0.5)
        check(f"seq2seq {tag}", o, out, 1e-5)
        p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}   # the restatement under autograd, fp64
        o64 = O.seq2seq_forward(p64, x.double(), last.double(), P, teacher.double(), coins, masks.double(), 0.5)
        (o64 * gw.double()).sum().backward()
        for k in grads:
            check(f"seq2seq {tag} grad {k}", p64[k].grad.float(), grads[k], 1e-4)
        cases.append(dict(tag=tag, dims=dict(C=C, H=H, layers=layers, P=P, O=No, ratio=ratio, R=R, T=T), params=p, x=x, last=last,
                          teacher=teacher, masks=masks, coins=coins, np_seed=seed + 3, out=out, gw=gw, grads=grads))
        seed += 10
    torch.save(cases, os.path.join(GOLD, "seq2seq.pt"))


if __name__ == "__main__":
    if len(sys.argv) > 1:                                     # python oracle/make_golden.py golden_seq2seq ...: selected fixtures only
        os.makedirs(GOLD, exist_ok=True)
        torch.set_num_threads(4)
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        main()
