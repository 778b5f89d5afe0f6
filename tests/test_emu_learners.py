"""CPU (host-emulated kernel build): the learners through the reference's own API vs fixtures recorded
from the real reference (tests/golden/*_train.pt, *_learn.pt; oracle/make_golden.py)."""
from types import SimpleNamespace

import pytest
import torch

from iplan_amd import _lib as L
from iplan_amd import synth
from tests.emu.emu_lib import get_emu_lib


@pytest.fixture(autouse=True)
def emu():
    L.use_library_for_tests(get_emu_lib())
    yield
    L.use_library_for_tests(None)


class RecLogger:
    def __init__(self):
        self.stats = {}

    def log_stat(self, k, v, t):
        self.stats[k] = float(v)


def max_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


def check_ippo_train(g, device):
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    from iplan_amd.learners.ippo_learner import IPPOLearner
    args = SimpleNamespace(**dict(g["args"], use_cuda=(device != "cpu")))
    scheme = synth.make_scheme(args)
    mac = DcntrlMAC(scheme, {"agents": args.n_agents}, args)
    for i in range(args.n_agents):
        mac.agents[i].load_state_dict(g["pre"]["actors"][i])
        mac.critics[i].load_state_dict(g["pre"]["critics"][i])
    log = RecLogger()
    learner = IPPOLearner(mac, scheme, log, args)
    E = g["fields"]["history"].shape[0]
    batch = synth.DictBatch(g["fields"], E, args.episode_limit + 1).to(device)
    learner.insert_episode_batch(batch)
    assert learner.buffers[0].can_sample()
    learner.train(0)
    assert not learner.buffers[0].can_sample()
    for i in range(args.n_agents):
        for name, mods in (("actors", mac.agents), ("critics", mac.critics)):
            sd = mods[i].state_dict()
            for k, ref in g["post"][name][i].items():
                assert max_rel(sd[k], ref) < 1e-5, (name, i, k, max_rel(sd[k], ref))
    for k, ref in g["stats"].items():
        got = log.stats[k]
        assert abs(got - ref) <= 1e-5 * max(1.0, abs(ref)), (k, got, ref)


@pytest.mark.parametrize("tag", ["ippo_train", "ippo_train_mpe", "ippo_train_tanh"])
def test_ippo_train_emulated(golden, tag):
    check_ippo_train(golden(tag), "cpu")


def check_prediction_learn(g, device):
    import numpy as np
    from iplan_amd.nova.prediction_policy import Prediction_policy
    args = SimpleNamespace(**dict(g["args"], use_cuda=(device != "cpu")))
    pol = Prediction_policy(args, RecLogger())
    nA, N, S, P = args.n_agents, args.max_vehicle_num, args.pred_batch_size, args.pred_length
    for i in range(nA):
        pol.pred_GAT[i].load_state_dict(g["pre"]["gat"][i])
        pol.pred_decoder[i].load_state_dict(g["pre"]["dec"][i])
    E = g["fields"]["history"].shape[0]
    batch = synth.DictBatch(g["fields"], E, args.episode_limit + 1).to(device)
    noise = torch.stack([x.reshape(S, N, N - 1, 2) for x in g["gumbel"]]).to(device)
    keep = torch.stack([torch.stack([g["dropout"][i * P + p].reshape(S * N, -1) for p in range(P)]) for i in range(nA)]).to(device)
    np.random.seed(g["np_seed"])
    losses = pol.learn(batch, 0, noise=noise, keep=keep.float().contiguous())
    for i in range(nA):
        assert abs(float(losses[i]) - g["losses"][i]) <= 1e-5 * max(1.0, abs(g["losses"][i])), (i, losses[i], g["losses"][i])
        for name, mods, arena in (("gat", pol.pred_GAT, pol.gat_arena), ("dec", pol.pred_decoder, pol.dec_arena)):
            for k, ref in g["clipped"][name][i].items():
                got = arena.grad_of(i, k).cpu()
                err = (got.double() - ref.double()).abs().max().item()
                assert err <= 1e-5 * ref.abs().max().item() + 1e-10, ("clipped grad", name, i, k, err)   # relative to the tensor's own scale
            sd = mods[i].state_dict()
            for k, ref in g["post"][name][i].items():
                assert max_rel(sd[k], ref) < 1e-6, ("post", name, i, k, max_rel(sd[k], ref))


def test_prediction_learn_emulated(golden):
    check_prediction_learn(golden("prediction_learn"), "cpu")


def check_behavior_learn(g, device):
    from iplan_amd.nova.stable_behavior_policy import Behavior_policy
    args = SimpleNamespace(**dict(g["args"], use_cuda=(device != "cpu")))
    pol = Behavior_policy(args, RecLogger())
    nA, N, Lw = args.n_agents, args.max_vehicle_num, args.max_history_len
    for i in range(nA):
        pol.behavior_encoder[i].load_state_dict(g["pre"]["enc"][i])
        pol.behavior_decoder[i].load_state_dict(g["pre"]["dec"][i])
    E = g["fields"]["history"].shape[0]
    J = args.episode_limit - 1 - Lw
    batch = synth.DictBatch(g["fields"], E, args.episode_limit + 1).to(device)
    keep = torch.stack([torch.stack(g["dropout"][i * J:(i + 1) * J]) for i in range(nA)])     # [nA, J, E*N, L, 64]
    bl, sl, tl = pol.learn(batch, 0, keep=keep.to(torch.uint8).contiguous().to(device))
    for i in range(nA):
        assert abs(float(bl[i]) - g["behavior_loss"][i]) <= 1e-5 * max(1.0, abs(g["behavior_loss"][i])), (i, bl[i])
        assert abs(float(sl[i]) - g["stability_loss"][i]) <= 1e-5 * max(1.0, abs(g["stability_loss"][i])), (i, sl[i])
        assert abs(float(tl[i]) - g["total_loss"][i]) <= 1e-5 * max(1.0, abs(g["total_loss"][i]))
        for name, mods, arena in (("enc", pol.behavior_encoder, pol.enc_arena), ("dec", pol.behavior_decoder, pol.dec_arena)):
            for k, ref in g["clipped"][name][i].items():
                got = arena.grad_of(i, k).cpu()
                err = (got.double() - ref.double()).abs().max().item()
                assert err <= 1e-5 * ref.abs().max().item() + 1e-9, ("clipped grad", name, i, k, err, ref.abs().max().item())
            sd = mods[i].state_dict()
            for k, ref in g["post"][name][i].items():
                assert max_rel(sd[k], ref) < 1e-6, ("post", name, i, k, max_rel(sd[k], ref))


def test_behavior_learn_emulated(golden):
    check_behavior_learn(golden("behavior_learn"), "cpu")


def test_behavior_learn_bptt_in_pieces_emulated(golden, monkeypatch):
    """The forward and the BPTT cut into window ranges (carries between the launches, weight gradients accumulated piece by
    piece -- the GPU path's pipelining) land on the same reference parameters (hard-update fixture: 4 windows, 3 pieces)."""
    monkeypatch.setenv("IPLAN_BEH_PIECES", "3")
    check_behavior_hard_learn(golden("behavior_hard_learn"), "cpu")


@pytest.mark.parametrize("tag", ["ippo_train_mpe", "ippo_train"])
def test_ippo_reference_shaped_methods_emulated(golden, tag):
    """compute_returns / generate_data / ppo_update (the reference's per-agent surface) reproduce the fused train():
    replaying the golden run agent by agent, epoch by epoch, lands on the reference's post-train parameters."""
    from tests.oracle_checks import check_ippo_reference_shaped_methods
    check_ippo_reference_shaped_methods(golden(tag), "cpu")


def check_behavior_hard_learn(g, device):
    from iplan_amd.nova.behavior_policy import Behavior_policy
    args = SimpleNamespace(**dict(g["args"], use_cuda=(device != "cpu")))
    pol = Behavior_policy(args, RecLogger())
    nA, Lw = args.n_agents, args.max_history_len
    for i in range(nA):
        pol.behavior_encoder[i].load_state_dict(g["pre"]["enc"][i])
        pol.behavior_decoder[i].load_state_dict(g["pre"]["dec"][i])
    E = g["fields"]["history"].shape[0]
    J = args.episode_limit // Lw - 1
    batch = synth.DictBatch(g["fields"], E, args.episode_limit + 1).to(device)
    keep = torch.stack([torch.stack(g["dropout"][i * J:(i + 1) * J]) for i in range(nA)])
    bl = pol.learn(batch, 0, keep=keep.to(torch.uint8).contiguous().to(device))
    for i in range(nA):
        assert abs(float(bl[i]) - g["behavior_loss"][i]) <= 1e-5 * max(1.0, abs(g["behavior_loss"][i])), (i, bl[i])
        for name, mods, arena in (("enc", pol.behavior_encoder, pol.enc_arena), ("dec", pol.behavior_decoder, pol.dec_arena)):
            for k, ref in g["clipped"][name][i].items():
                err = (arena.grad_of(i, k).cpu().double() - ref.double()).abs().max().item()
                assert err <= 1e-5 * ref.abs().max().item() + 1e-9, ("clipped grad", name, i, k, err)
            sd = mods[i].state_dict()
            for k, ref in g["post"][name][i].items():
                assert max_rel(sd[k], ref) < 1e-6, ("post", name, i, k)


def test_behavior_hard_learn_emulated(golden):
    check_behavior_hard_learn(golden("behavior_hard_learn"), "cpu")


def check_behavior_fc_learn(g, device):
    """iPLAN-FC ablation: rollout latent, loss, clipped gradients and post-Adam parameters vs the reference."""
    from iplan_amd.nova.behavior_FC_policy import Behavior_policy
    args = SimpleNamespace(**dict(g["args"], use_cuda=(device != "cpu")))
    pol = Behavior_policy(args, RecLogger())
    nA = args.n_agents
    for i in range(nA):
        pol.behavior_encoder[i].load_state_dict(g["pre"]["enc"][i])
        pol.behavior_decoder[i].load_state_dict(g["pre"]["dec"][i])
    lat, hid = pol.latent_update(g["window"].numpy(), None, None)
    assert hid is None and max_rel(torch.as_tensor(lat), g["latent"]) < 1e-5
    E = g["fields"]["history"].shape[0]
    batch = synth.DictBatch(g["fields"], E, args.episode_limit + 1).to(device)
    bl, stab, total = pol.learn(batch, 0)
    assert stab == [] and len(total) == nA
    changed = 0.0
    for i in range(nA):
        assert abs(float(bl[i]) - g["behavior_loss"][i]) <= 1e-5 * max(1.0, abs(g["behavior_loss"][i])), (i, bl[i])
        for name, mods, arena in (("enc", pol.behavior_encoder, pol.enc_arena), ("dec", pol.behavior_decoder, pol.dec_arena)):
            for k, ref in g["clipped"][name][i].items():
                err = (arena.grad_of(i, k).cpu().double() - ref.double()).abs().max().item()
                assert err <= 1e-5 * ref.abs().max().item() + 1e-9, ("clipped grad", name, i, k, err)
            sd = mods[i].state_dict()
            for k, ref in g["post"][name][i].items():
                assert max_rel(sd[k], ref) < 1e-6, ("post", name, i, k)
                changed = max(changed, (ref - g["pre"][name][i][k]).abs().max().item())
    assert changed > 0                                           # the fixture really moved the parameters
    # the module-level forwards (nova/behavior_FC_net.py) on the kernels
    from iplan_amd.nova.behavior_FC_net import Encoder_3FC
    enc = Encoder_3FC(args.obs_shape_single * args.max_history_len, args.encoder_rnn_dim, args.latent_dim)
    enc.load_state_dict(g["pre"]["enc"][0])
    x = g["window"][:, 0].reshape(E, args.max_vehicle_num, -1).to(device)
    assert max_rel(enc(x).cpu(), g["latent"][:, 0]) < 1e-5


def test_behavior_fc_learn_emulated(golden):
    check_behavior_fc_learn(golden("behavior_fc_learn"), "cpu")


def check_ippo_train_vs_oracle(g, device, mutate=None, tol=1e-5, **arg_overrides):
    """IPPOLearner.train against the oracle's per-agent PPO replay (oracle.ppo_train_agent) on the fixture's episode fields,
    optionally mutated -- covers buffer contents the recorded reference run does not (unfilled trailing steps ...)."""
    import copy
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    from iplan_amd.learners.ippo_learner import IPPOLearner
    from oracle import iplan_oracle as O
    args = SimpleNamespace(**dict(g["args"], use_cuda=(device != "cpu"), **arg_overrides))
    fields = copy.deepcopy(g["fields"])
    if mutate is not None:
        mutate(fields, args)
    scheme = synth.make_scheme(args)
    mac = DcntrlMAC(scheme, {"agents": args.n_agents}, args)
    for i in range(args.n_agents):
        mac.agents[i].load_state_dict(g["pre"]["actors"][i])
        mac.critics[i].load_state_dict(g["pre"]["critics"][i])
    learner = IPPOLearner(mac, scheme, RecLogger(), args)
    E = fields["history"].shape[0]
    learner.insert_episode_batch(synth.DictBatch(fields, E, args.episode_limit + 1).to(device))
    learner.train(0)
    worst = 0.0
    for i in range(args.n_agents):
        ap = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in g["pre"]["actors"][i].items()}
        cp = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in g["pre"]["critics"][i].items()}
        O.ppo_train_agent(i, ap, cp, fields, args)
        for name, ref_p, mods in (("actor", ap, mac.agents), ("critic", cp, mac.critics)):
            sd = mods[i].state_dict()
            for k, ref in ref_p.items():
                e = max_rel(sd[k], ref.detach())
                worst = max(worst, e)
                assert e < tol, (name, i, k, e)
    return worst


def unfilled_tail(fields, args):
    """What ippo_parallel_runner.py:212-214 leaves behind when every env terminates early: the trailing steps were never
    written -- ``actions`` = 0, ``actions_onehot`` = all zeros (OneHot only runs on updated slices), terminated = 0."""
    for k in ("actions", "actions_onehot", "terminated", "reward"):
        fields[k][:, -3:] = 0
    fields["actions"][:, 0] = 2               # make sure index 0 is not what a wrong reconstruction would need


def test_ippo_train_unfilled_trailing_steps_emulated(golden):
    check_ippo_train_vs_oracle(golden("ippo_train"), "cpu", mutate=unfilled_tail)


def test_per_agent_buffer_surface_emulated(golden):
    """The reference-shaped per-agent loop (buffers[i].get_batch() ... buffers[i].clear_buffer(), agent after agent,
    separated_buffer.py:39-50) sees every agent's data; the shared store is released when the last agent has cleared."""
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    from iplan_amd.learners.ippo_learner import IPPOLearner
    g = golden("ippo_train_mpe")
    args = SimpleNamespace(**dict(g["args"], use_cuda=False))
    scheme = synth.make_scheme(args)
    learner = IPPOLearner(DcntrlMAC(scheme, {"agents": args.n_agents}, args), scheme, RecLogger(), args)
    f = dict(g["fields"])
    f.pop("state")                                                   # optional field: get_batch must not require it
    learner.insert_episode_batch(synth.DictBatch(f, f["history"].shape[0], args.episode_limit + 1))
    for i in range(args.n_agents):
        b = learner.buffers[i].get_batch()
        assert b is not None and "state" not in b and torch.equal(b["history"], f["history"][:, :, i])
        learner.buffers[i].clear_buffer()
        assert learner.buffers[i].get_batch() is None
    assert learner.store.count == 0


def test_optimizer_load_resets_state_emulated():
    """load_state_dict into an already-trained optimiser: parameters absent from the file get zero moments, an empty state
    resets the step count (torch.optim.Adam semantics); moment arenas live on the arena object."""
    from iplan_amd.arena import ParamArena
    from iplan_amd.optim import FusedAdam
    torch.manual_seed(0)
    mods = [torch.nn.Linear(4, 3)]
    arena = ParamArena(mods, "cpu")
    opt = FusedAdam([(arena, 0)], lr=1e-2, eps=1e-5)
    fresh = opt.state_dict()
    assert fresh["state"] == {}
    for p in mods[0].parameters():
        p.grad.copy_(torch.randn_like(p))
    opt.step(max_norm=1.0)
    sd = opt.state_dict()
    assert set(sd["state"].keys()) == {0, 1} and opt._steps == 1
    partial = {"state": {0: sd["state"][0]}, "param_groups": sd["param_groups"]}
    opt.load_state_dict(partial)
    m, v = arena._adam_moments
    bias = arena.ranges(["bias"])[0]
    assert float(m[0, bias[0]:bias[0] + bias[1]].abs().max()) == 0.0 and float(v[0, bias[0]:bias[0] + bias[1]].abs().max()) == 0.0
    assert set(opt.state_dict()["state"].keys()) == {0}
    opt.load_state_dict(fresh)
    assert opt._steps == 0 and float(m.abs().max()) == 0.0


# ---- oracle-driven checks at arbitrary sizes (tests/oracle_checks.py): small sizes here, BASELINE config 2 / 3 / 5 sizes on the GPU
def _small(**kw):
    from iplan_amd.config import default_args
    base = dict(use_cuda=False, max_vehicle_num=5, n_agents=2, episode_limit=9, batch_size_run=3, max_history_len=3,
                pred_batch_size=6, pred_length=2, buffer_size=4, batch_size=3, ppo_epoch=2)
    base.update(kw)
    return default_args("highway", **base)


def test_behavior_learn_vs_oracle_emulated():
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    w = check_behavior_learn_vs_oracle(_small(), 3, "cpu", seed=3)
    assert w["grad"] < 1e-5, w


@pytest.mark.parametrize("cfg", ["iplan", "gat_only"])
def test_prediction_learn_vs_oracle_emulated(cfg):
    from tests.oracle_checks import check_prediction_learn_vs_oracle
    kw = {} if cfg == "iplan" else dict(Behavior_enable=False)
    check_prediction_learn_vs_oracle(_small(**kw), 3, "cpu", seed=4, gate_e32_factor=4.0)     # (small case: see the check's docstring)


@pytest.mark.parametrize("cfg", ["iplan", "gat_only", "plain"])
def test_ppo_train_vs_oracle_emulated(cfg):
    """config 3 (F = N(d+A+Z)+10), config 2 (Behaviour off: F = N(d+A)+10) and config 1 (both off) feature layouts"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    kw = dict(iplan={}, gat_only=dict(Behavior_enable=False), plain=dict(Behavior_enable=False, GAT_enable=False, GAT_use_behavior=False))[cfg]
    check_ppo_train_vs_oracle(_small(**kw), "cpu", seed=6)


@pytest.mark.parametrize("decay", [False, True])
def test_ppo_train_15_epochs_vs_oracle_emulated(decay):
    """the shipped ppo_epoch = 15 (config/algs/ippo.yaml:6; learners/ippo_learner.py:286-303): gradients at the learner's own
    parameters in front of optimiser steps 3, 8 and 15 vs the fp64 oracle, one fp64 Adam step from each of those states, and EVERY one
    of the 15 Adam updates replayed in fp64 from its own state (bias correction at t = 1 .. 15); with the linear lr decay hook
    on (use_linear_lr_decay, t_env = 0.4 t_max: lr and critic_lr x 0.6) in the second case"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    kw = dict(use_linear_lr_decay=True, t_max=1000) if decay else {}
    w = check_ppo_train_vs_oracle(_small(ppo_epoch=15, **kw), "cpu", seed=6, mid_probes=(2, 7), adam_replay=True, t_env=400 if decay else 0)
    assert w["adam_replay_updates"] == 15 * 2 * 2 and "mid_grad" in w, w


def test_ppo_loss_in_parts_vs_oracle_emulated(monkeypatch):
    """iplan_ppo_loss with the rows of an agent dealt to several workgroups (IplanPpoLossArgs.n_parts: what 22 950 rows get) -- ragged
    last range, mask sum handed in; gradients and the logged statistics (sum of the ranges' shares) against the oracle"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    monkeypatch.setenv("IPLAN_PPO_LOSS_PARTS", "4")
    check_ppo_train_vs_oracle(_small(batch_size_run=5, buffer_size=6, batch_size=5), "cpu", seed=17, check_stats=True)


def test_ppo_train_wide_tail_workgroups_vs_oracle_emulated(monkeypatch):
    """the PPO epoch's forward tail in its 16-wave workgroup form (what a full 22 950-row batch gets; small batches default to 8 waves)"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    monkeypatch.setenv("IPLAN_AC_PRE_WAVES", "16")
    check_ppo_train_vs_oracle(_small(batch_size_run=5, buffer_size=6, batch_size=5), "cpu", seed=16)


def test_ppo_train_tanh_vs_oracle_emulated():
    """args.use_ReLU off (utils/mappo_utils/mlp.py:10): tanh in fc1 / fc2 of actor and critic, forward and backward tail"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    w = check_ppo_train_vs_oracle(_small(use_ReLU=False), "cpu", seed=6)
    assert w.get("relu_branches_from_hint", 0) == 0, w


def test_gat_fwd_bwd_vs_oracle_emulated():
    """GAT forward + backward at config 5's input width (D = 128) and a ragged entity count, vs the fp64 oracle"""
    from tests.oracle_checks import check_gat_fwd_bwd_vs_oracle
    check_gat_fwd_bwd_vs_oracle(B=2, N=9, D=128, device="cpu", seed=8, gate_e32_factor=4.0)  # (small case: few gates, see check_prediction_learn_vs_oracle)


def test_behavior_learn_env_chunks_vs_oracle_emulated(monkeypatch):
    """Behavior_policy.learn on env chunks (config 4's 256 envs on one GPU do not fit one launch's BPTT records): the chunk
    gradients accumulate in the arenas under the all-env window normalisers -- same result as the oracle on the whole batch."""
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    monkeypatch.setenv("IPLAN_BEH_ENV_CHUNK", "2")
    w = check_behavior_learn_vs_oracle(_small(), 5, "cpu", seed=13)
    assert w["grad"] < 1e-5, w
    # ... and against the oracle evaluated in env shares of ANOTHER size, fp64 only (the form of the config-4 GPU test)
    w = check_behavior_learn_vs_oracle(_small(), 5, "cpu", seed=13, oracle_env_chunk=3, fp32_oracle=False)
    assert w["grad"] < 1e-5, w


def test_behavior_learn_deferred_decoder_vs_oracle_emulated():
    """learn(defer_decoder=True): the decoder's weight gradients / clip / Adam leave the call (side stream on the GPU, in line
    here) -- same gradients and post-step parameters as the oracle, for two consecutive calls"""
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    w = check_behavior_learn_vs_oracle(_small(), 3, "cpu", seed=19, learn_kwargs=dict(defer_decoder=True))
    assert w["grad"] < 1e-5 and w["post"] < 1e-6, w


def test_behavior_learn_deferred_twice_equals_inline_emulated():
    from tests.oracle_checks import check_deferred_equals_inline
    check_deferred_equals_inline(_small(), 3, "cpu")


def test_behavior_learn_with_stability_penalty_vs_oracle_emulated():
    """behavior_variation_penalty != 0 (nova/stable_behavior_policy.py:238-246): the stability term is differentiated too"""
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    w = check_behavior_learn_vs_oracle(_small(behavior_variation_penalty=0.3, thres_small_variation=0.05), 3, "cpu", seed=17)
    assert w["grad"] < 1e-5, w


def test_adam_weight_decay_emulated():
    """weight_decay != 0 (torch.optim.Adam's L2 form, applied after the clip) against torch.optim.Adam itself"""
    from iplan_amd.arena import ParamArena
    from iplan_amd.optim import FusedAdam
    torch.manual_seed(0)
    mods = [torch.nn.Linear(6, 4)]
    ref = torch.nn.Linear(6, 4)
    ref.load_state_dict(mods[0].state_dict())
    arena = ParamArena(mods, "cpu")
    opt = FusedAdam([(arena, 0)], lr=1e-2, eps=1e-5, weight_decay=0.05)
    topt = torch.optim.Adam(ref.parameters(), lr=1e-2, eps=1e-5, weight_decay=0.05)
    for step in range(3):
        grads = [torch.randn_like(p) * 3 for p in ref.parameters()]
        for p, q, gq in zip(mods[0].parameters(), ref.parameters(), grads):
            p.grad.copy_(gq)
            q.grad = gq.clone()
        opt.step(max_norm=1.0)
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        topt.step()
        for p, q in zip(mods[0].parameters(), ref.parameters()):
            assert max_rel(p.detach(), q.detach()) < 1e-6


def test_ppo_minibatches_vs_oracle_emulated():
    """num_mini_batch > 1 (generate_data's randperm split, one optimiser step per minibatch): same permutations as the
    reference draws them, post-train parameters and last-step gradients vs the oracle"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    check_ppo_train_vs_oracle(_small(num_mini_batch=3, ppo_epoch=2, episode_limit=10, batch_size=3), "cpu", seed=31)


@pytest.mark.parametrize("flags", [
    dict(use_huber_loss=False), dict(use_clipped_value_loss=False), dict(use_value_active_masks=False),
    dict(use_policy_active_masks=False), dict(use_gae=False),
    dict(use_huber_loss=False, use_clipped_value_loss=False, use_value_active_masks=False, use_policy_active_masks=False, use_gae=False)])
def test_ppo_loss_switches_vs_oracle_emulated(flags):
    """the PPO loss switches of config/algs/ippo.yaml away from their shipped values (learners/ippo_learner.py:142-157,
    190-196, 353-362): MSE instead of Huber, no value clipping, plain means instead of active masks, no GAE"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    # Same bound as everywhere (tests/oracle_checks.py: E32_FACTOR, plus the data-derived conditioning term of the two row-sum
    # tensors).  All switches off at seed 41 is the draw where it matters: the critic's v_out.bias / rnn.norm.bias gradients are
    # means of v - return that cancel to 1 / 300 over the 27 rows; seed 42 is an ordinary draw of the same case.
    w = check_ppo_train_vs_oracle(_small(ppo_epoch=2, **flags), "cpu", seed=41)
    if len(flags) > 1:
        assert w["value_grad_row_sum_cond"] > 50, w                                 # (the ill-conditioned draw really is one)
        w2 = check_ppo_train_vs_oracle(_small(ppo_epoch=2, **flags), "cpu", seed=42)
        assert w2["grad"] < 1e-5, w2


def test_behavior_learn_decoder_forward_both_forms_emulated(monkeypatch):
    """the decoder forward's two forms -- the default second form (split-bf16 products, register-resident weights, eight
    role-split waves per tile set handing data over through LDS counters) and the first form (IPLAN_DEC_FWD_V1=1: fp32 MFMA,
    weights in LDS, four quarter-waves per tile) -- give the same loss and gradients, both within 1e-5 of the fp64 oracle"""
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    w2 = check_behavior_learn_vs_oracle(_small(max_vehicle_num=7, episode_limit=14, max_history_len=4), 3, "cpu", seed=3)
    monkeypatch.setenv("IPLAN_DEC_FWD_V1", "1")
    w1 = check_behavior_learn_vs_oracle(_small(max_vehicle_num=7, episode_limit=14, max_history_len=4), 3, "cpu", seed=3)
    monkeypatch.delenv("IPLAN_DEC_FWD_V1")
    for w in (w1, w2):
        assert w["grad"] < 1e-5 and w["loss"] < 1e-5, (w1, w2)
    assert abs(w2["grad"] - w1["grad"]) < 1e-6, (w1, w2)


def test_behavior_learn_encoder_both_forms_emulated(monkeypatch):
    """the behaviour encoder's forward and BPTT in their two forms -- the default split-bf16 form (round 4: W_ih u / W_hh h and
    the two backward-data products on the bf16 matrix cores, fp32-exact) and the fp32-MFMA form (IPLAN_ENC_FP32=1) -- both
    within 1e-5 of the fp64 oracle, at a ragged size (partial tile) and through BPTT window pieces"""
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    kw = dict(max_vehicle_num=7, episode_limit=14, max_history_len=4)
    w2 = check_behavior_learn_vs_oracle(_small(**kw), 3, "cpu", seed=5)
    monkeypatch.setenv("IPLAN_ENC_FP32", "1")
    w1 = check_behavior_learn_vs_oracle(_small(**kw), 3, "cpu", seed=5)
    monkeypatch.delenv("IPLAN_ENC_FP32")
    for w in (w1, w2):
        assert w["grad"] < 1e-5 and w["loss"] < 1e-5, (w1, w2)
    assert abs(w2["grad"] - w1["grad"]) < 1e-6, (w1, w2)
    monkeypatch.setenv("IPLAN_BEH_PIECES", "3")
    w3 = check_behavior_learn_vs_oracle(_small(**kw), 3, "cpu", seed=5)
    assert w3["grad"] < 1e-5, w3


def test_behavior_learn_decoder_bptt_both_forms_emulated(monkeypatch):
    """the decoder BPTT's two forms -- the default second form (round 4: split-bf16 backward-data products, register-resident
    weight pieces, eight role-split waves per tile set, LDS-counter hand-offs) and the first form (IPLAN_DEC_BWD_V1=1: fp32 MFMA,
    weights in LDS, four quarter-waves per tile) -- both within 1e-5 of the fp64 oracle: a ragged size (partial and missing tiles
    in a workgroup), BPTT window pieces (carry of d loss / d h), the stability penalty, more than three tiles per net"""
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    kw = dict(max_vehicle_num=7, episode_limit=14, max_history_len=4)
    for extra, E, seed in ((dict(), 3, 3), (dict(behavior_variation_penalty=0.3, thres_small_variation=0.05), 3, 17),
                           (dict(max_vehicle_num=13), 9, 7)):                     # 117 chains = 8 tiles (3 workgroups, last: 2 tiles, ragged)
        args = dict(kw, **extra)
        w2 = check_behavior_learn_vs_oracle(_small(**args), E, "cpu", seed=seed)
        monkeypatch.setenv("IPLAN_DEC_BWD_V1", "1")
        w1 = check_behavior_learn_vs_oracle(_small(**args), E, "cpu", seed=seed)
        monkeypatch.delenv("IPLAN_DEC_BWD_V1")
        monkeypatch.setenv("IPLAN_DEC_THIN_ROWS", "1")               # second form, thin weight gradients from row gradients by iplan_wgrad
        w2r = check_behavior_learn_vs_oracle(_small(**args), E, "cpu", seed=seed)
        monkeypatch.delenv("IPLAN_DEC_THIN_ROWS")
        for w in (w1, w2, w2r):
            assert w["grad"] < 1e-5 and w["loss"] < 1e-5, (w1, w2, w2r)
        assert abs(w2r["grad"] - w1["grad"]) < 1e-6, (w1, w2r)          # same contraction kernels: the BPTT forms agree closely
        assert abs(w2["grad"] - w1["grad"]) < 3e-6, (w1, w2)            # (in-kernel thin gradients: another summation order)
    monkeypatch.setenv("IPLAN_BEH_PIECES", "3")
    w3 = check_behavior_learn_vs_oracle(_small(**kw), 3, "cpu", seed=3)
    assert w3["grad"] < 1e-5, w3


OTHER_DIMS = dict(obs_shape_single=7, latent_dim=5, max_history_len=4, episode_limit=12, n_actions=4, n_agents=3, max_vehicle_num=6, pred_length=3)


def check_other_runtime_dims(device):
    """Every RUN-TIME dimension of the path off its shipped value at once -- entity features d = 7 (5), latent Z = 5 (8), window
    L = 4 (10), 4 actions (5), 3 agents (5), prediction horizon 3 (5), N = 6, T = 12: the three learners against the oracle.  (What is
    fixed at compile time are the hidden widths 64 / 32 / 32 / 64, one GRU layer, and the MLP depth: DESIGN.md section 7.)"""
    from tests.oracle_checks import check_behavior_learn_vs_oracle, check_ppo_train_vs_oracle, check_prediction_learn_vs_oracle
    out = dict(behaviour=check_behavior_learn_vs_oracle(_small(**OTHER_DIMS), 3, device, seed=3),
               prediction=check_prediction_learn_vs_oracle(_small(**OTHER_DIMS), 3, device, seed=4, gate_e32_factor=4.0),
               ppo=check_ppo_train_vs_oracle(_small(**OTHER_DIMS), device, seed=6))
    return out


def test_other_runtime_dims_vs_oracle_emulated():
    check_other_runtime_dims("cpu")
