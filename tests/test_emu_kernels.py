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


def _args_from(g, **kw):
    from types import SimpleNamespace
    d = dict(g["args"])
    d.update(kw)
    return SimpleNamespace(**d)


class _NullLogger:
    def log_stat(self, *a, **k):
        pass


def test_encoder_forward_emulated(golden):
    from iplan_amd.nova.behavior_net import EncoderRNN
    g = golden("encoder")
    net = EncoderRNN(5, 32, 8, 1)
    net.load_state_dict(g["params"])
    with torch.no_grad():
        _, hL, lat = net(g["x"], g["h0"].unsqueeze(0))
    assert rel_err(hL[0], g["hL"]) < 1e-5
    assert rel_err(lat, g["latent"]) < 1e-5


def check_rollout_step(g, device):
    """Shared by the emulated (CPU) and the GPU test: drop-in API vs reference outputs."""
    import numpy as np
    from iplan_amd import synth
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    from iplan_amd.nova.prediction_policy import Prediction_policy
    from iplan_amd.nova.stable_behavior_policy import Behavior_policy
    args = _args_from(g, use_cuda=(device != "cpu"))
    pred = Prediction_policy(args, _NullLogger())
    beh = Behavior_policy(args, _NullLogger())
    mac = DcntrlMAC(synth.make_scheme(args), {"agents": args.n_agents}, args)
    for i in range(args.n_agents):
        pred.pred_GAT[i].load_state_dict(g["gat"][i])
        beh.behavior_encoder[i].load_state_dict(g["enc"][i])
        mac.agents[i].load_state_dict(g["actors"][i])
        mac.critics[i].load_state_dict(g["critics"][i])
    nA, E = args.n_agents, g["att0"].shape[0]
    N = args.max_vehicle_num
    noise = torch.stack([x.reshape(E, N, N - 1, 2) for x in g["gumbel"]]).to(device)
    att1 = pred.GAT_latent_update(g["hist_single"].numpy(), g["att0"].numpy(), g["lat0"].numpy(), noise=noise)
    assert isinstance(att1, np.ndarray) and att1.dtype == np.float32
    assert rel_err(torch.as_tensor(att1), g["att1"]) < 1e-5
    lat1, eh1 = beh.latent_update(g["window"].numpy(), g["eh0"].numpy(), g["lat0"].numpy())
    assert isinstance(lat1, np.ndarray) and torch.is_tensor(eh1)
    assert rel_err(torch.as_tensor(lat1), g["lat1"]) < 1e-5
    assert rel_err(eh1.cpu(), g["eh1"]) < 1e-5
    batch = synth.DictBatch(g["fields"], E, args.episode_limit + 1).to(device)
    t = g["t_ep"]
    for name, tt, test_mode, q in (("det", t, True, None), ("smp", t, False, g["q"].to(device)), ("t0", 0, True, None)):
        vals, acts, logps, ha, hc = mac.select_actions_ippo(batch, tt, test_mode=test_mode, q_noise=q)
        ref = g[name]
        assert vals.shape == (E, nA) and acts.shape == (E, nA) and acts.dtype == np.int64
        assert np.array_equal(acts, ref["actions"].numpy()), name
        assert rel_err(torch.as_tensor(vals), ref["values"]) < 1e-5, name
        for i in range(nA):
            assert logps[i].shape == (E, 1)
            assert rel_err(logps[i].cpu(), ref["logp"][i]) < 1e-5, (name, i)
        if name == "det":
            assert ha.shape == (1, E, nA, 64)
            assert rel_err(torch.as_tensor(ha), ref["h_actor"]) < 1e-5
            assert rel_err(torch.as_tensor(hc), ref["h_critic"]) < 1e-5
    # Pre-packed fc1 operands (iplan_ac_pack_fc1, incl. the lazily packed W gamma / W beta of the folded form) against the
    # arenas read in place (what a C-ABI caller without a pack gets) -- also right after the weights change under the cache.
    import os

    def both():
        packed = mac.select_actions_ippo(batch, t, test_mode=True)
        os.environ["IPLAN_NO_FC1_PACK"] = "1"
        try:
            plain = mac.select_actions_ippo(batch, t, test_mode=True)
        finally:
            del os.environ["IPLAN_NO_FC1_PACK"]
        return packed, plain

    first, _ = both()
    for arena in (mac.actor_arena, mac.critic_arena):
        arena.data.mul_(0.75)                                   # torch-side write: bumps data._version
    (v1, a1, lp1, ha1, hc1), (v2, a2, lp2, ha2, hc2) = both()
    assert rel_err(torch.as_tensor(v1), torch.as_tensor(v2)) < 2e-6 and np.array_equal(a1, a2)
    assert rel_err(torch.as_tensor(ha1), torch.as_tensor(ha2)) < 2e-6 and rel_err(torch.as_tensor(hc1), torch.as_tensor(hc2)) < 2e-6
    assert rel_err(torch.as_tensor(first[0]), torch.as_tensor(v1)) > 1e-3      # (the weights did change)


def test_rollout_step_emulated(golden):
    check_rollout_step(golden("rollout_step"), "cpu")


def test_clip_adam_emulated():
    from iplan_amd.arena import ParamArena
    from iplan_amd.optim import FusedAdam, step_all
    from oracle import iplan_oracle as O
    torch.manual_seed(0)
    mods = [torch.nn.Linear(7, 5) for _ in range(3)]
    arena = ParamArena(mods, "cpu")
    opts = [FusedAdam([(arena, i)], lr=1e-2, eps=1e-5) for i in range(3)]
    ref_p = [[p.detach().clone() for p in m.parameters()] for m in mods]
    ref_m = [[torch.zeros_like(p) for p in ps] for ps in ref_p]
    ref_v = [[torch.zeros_like(p) for p in ps] for ps in ref_p]
    for step in (1, 2, 3):
        grads = [[torch.randn_like(p) * (5.0 if i == 1 else 0.1) for p in ps] for i, ps in enumerate(ref_p)]
        for i, m in enumerate(mods):
            for p, gq in zip(m.parameters(), grads[i]):
                p.grad.copy_(gq)
        if step < 3:
            step_all(opts, 1.0)
        else:
            for o in opts:
                o.step(max_norm=1.0)
        for i in range(3):
            gl = [x.clone() for x in grads[i]]
            O.clip_grad_norm(gl, 1.0)
            for k in range(len(gl)):
                O.adam_step(ref_p[i][k], gl[k], ref_m[i][k], ref_v[i][k], step, 1e-2, 1e-5)
            for p, r in zip(mods[i].parameters(), ref_p[i]):
                assert rel_err(p.detach(), r) < 1e-6
    sd = opts[0].state_dict()
    assert set(sd["state"].keys()) == {0, 1} and sd["state"][0]["exp_avg"].shape == (5, 7)


def test_harness_rollout_emulated():
    """The synthetic training loop (device-resident stand-in for ParallelRunner) runs end to end."""
    from iplan_amd.config import default_args
    from iplan_amd.harness import SyntheticLoop
    args = default_args("highway", use_cuda=False, max_vehicle_num=5, n_agents=2, episode_limit=3, batch_size_run=2)
    loop = SyntheticLoop(args, 2, seed=0, device="cpu")
    batch = loop.rollout()
    assert torch.isfinite(batch["attention_latent"]).all()
    assert batch["attention_latent"][:, 1:].abs().sum() > 0
    # latent starts at 0 and moves 10 % towards a simplex point every step: row sums are 1 - 0.9^t
    sums = batch["behavior_latent"].sum(-1)
    for t in range(1, 4):
        assert (sums[:, t] - (1 - 0.9 ** t)).abs().max() < 1e-5
    assert batch["actions"].max() < args.n_actions


def test_fused_gat_encoder_launch_equals_two_launches_emulated(monkeypatch):
    """iplan_gat_enc_fwd (one launch per rollout step for GAT_latent_update + the encoder's latent_update) writes bit for bit what
    the two separate launches write"""
    from iplan_amd.config import default_args
    from iplan_amd.harness import SyntheticLoop
    args = default_args("highway", use_cuda=False, max_vehicle_num=19, n_agents=2, episode_limit=3, batch_size_run=9)   # 171 rows: ragged encoder tiles

    def run(no_fuse):
        if no_fuse:
            monkeypatch.setenv("IPLAN_NO_FUSE_ENC", "1")
        else:
            monkeypatch.delenv("IPLAN_NO_FUSE_ENC", raising=False)
        loop = SyntheticLoop(args, 9, seed=3, device="cpu")
        torch.manual_seed(11)
        b = loop.rollout()
        return {k: b[k].clone() for k in ("attention_latent", "behavior_latent", "actions", "rnn_states_actors")}

    fused, plain = run(False), run(True)
    for k in fused:
        assert torch.equal(fused[k], plain[k]), k
    assert fused["behavior_latent"][:, 1:].abs().sum() > 0 and fused["attention_latent"][:, 1:].abs().sum() > 0


def test_harness_full_cycle_emulated(capsys):
    """rollout -> insert -> behaviour learn -> prediction learn -> PPO train, end to end (tiny dims)."""
    from iplan_amd.config import default_args
    from iplan_amd.harness import SyntheticLoop
    args = default_args("highway", use_cuda=False, max_vehicle_num=3, n_agents=2, episode_limit=10, batch_size_run=2,
                        buffer_size=2, batch_size=1, ppo_epoch=2, pred_batch_size=4, max_history_len=3)
    loop = SyntheticLoop(args, 2, seed=0, device="cpu")
    before = [p.detach().clone() for p in loop.mac.agents[0].parameters()]
    enc_before = loop.behavior.enc_arena.data.clone()
    gat_before = loop.prediction.gat_arena.data.clone()
    for o in loop.obs_sets:
        # Highway polarity: the prediction / behaviour loss mask IS `terminated`, PPO's is 1 - terminated
        o["terminated"].copy_((torch.rand(o["terminated"].shape) < 0.5).to(torch.uint8))
    n = loop.cycle()
    assert n == 2 * 10
    assert loop.learner.last_train_info is not None and not loop.learner.buffers[0].can_sample()
    assert any((a - b).abs().max() > 0 for a, b in zip(before, loop.mac.agents[0].parameters()))
    assert (loop.behavior.enc_arena.data - enc_before).abs().max() > 0
    assert (loop.prediction.gat_arena.data - gat_before).abs().max() > 0
    for arena in (loop.mac.actor_arena, loop.mac.critic_arena, loop.behavior.enc_arena, loop.behavior.dec_arena,
                  loop.prediction.gat_arena, loop.prediction.dec_arena):
        assert torch.isfinite(arena.data).all()


def check_decoder_modules(golden, device):
    """Prediction_Decoder.forward and Behavior_Latent_Decoder.forward as plain modules vs the reference outputs."""
    from iplan_amd.nova.behavior_net import Behavior_Latent_Decoder
    from iplan_amd.nova.prediction_net import Prediction_Decoder
    g = golden("pred_decoder")
    net = Prediction_Decoder(5, 32, 1, 5, 5, dropout=0.1, teacher_forcing_ratio=0)
    net.load_state_dict(g["params"])
    keep = g["masks"].reshape(5, -1, 32).unsqueeze(0).float().contiguous().to(device)
    with torch.no_grad():
        pred = net(g["last"].to(device), torch.zeros_like(g["pred"]).to(device), g["hidden"].to(device), keep=keep)
    assert rel_err(pred.cpu(), g["pred"]) < 1e-5
    g = golden("decoder")
    dec = Behavior_Latent_Decoder(13, 64, 1, 5, dropout=0.1)
    dec.load_state_dict(g["params"])
    E_N, Lw, _ = g["dec_in"].shape
    keep = g["mask"].reshape(1, 1, E_N, Lw, 64).to(torch.uint8).contiguous().to(device)
    with torch.no_grad():
        y, hT = dec(g["curr"].to(device), g["latent"].to(device), g["h0"].unsqueeze(0).to(device), keep=keep)
    assert rel_err(y.cpu(), g["y"]) < 1e-5 and rel_err(hT[0].cpu(), g["hT"]) < 1e-5


def test_decoder_module_forwards_emulated(golden):
    check_decoder_modules(golden, "cpu")


def check_reference_checkpoint(golden, device, tmp_path):
    """The reference's own shipped checkpoint files (trained IPPO actor / critic / Adam state, input dim 85) load
    into the mirrored classes key for key, give the reference's outputs, and round-trip through save/load."""
    from iplan_amd.config import default_args
    from iplan_amd.modules.agents.ippo_actor import R_Actor
    from iplan_amd.modules.critics.ippo_critic import R_Critic
    from iplan_amd.arena import ParamArena
    from iplan_amd.optim import FusedAdam
    g = golden("checkpoint_fixture")
    args = default_args("highway", use_cuda=(device != "cpu"))
    actor, critic = R_Actor(85, args), R_Critic(85, args)
    assert list(actor.state_dict().keys()) == list(g["actor"].keys())
    assert list(critic.state_dict().keys()) == list(g["critic"].keys())
    assert str(actor.load_state_dict(g["actor"])) == "<All keys matched successfully>"
    assert str(critic.load_state_dict(g["critic"])) == "<All keys matched successfully>"
    with torch.no_grad():
        act, logp, h_a = actor(g["x"].to(device), g["h"].to(device), g["avail"].to(device), deterministic=True)
        val, h_c = critic(g["x"].to(device), g["h"].to(device))
    assert torch.equal(act.cpu(), g["actions"])
    assert rel_err(logp.cpu(), g["logp"]) < 1e-5 and rel_err(h_a.cpu(), g["h_actor"]) < 1e-5
    assert rel_err(val.cpu(), g["values"]) < 1e-5 and rel_err(h_c.cpu(), g["h_critic"]) < 1e-5
    # optimiser state in torch.optim.Adam's file format
    arena = ParamArena([actor], device)
    opt = FusedAdam([(arena, 0)], lr=args.lr, eps=args.optim_eps)
    opt.load_state_dict(g["actor_opt"])
    sd = opt.state_dict()
    assert set(sd["state"].keys()) == set(g["actor_opt"]["state"].keys())
    for k, st in g["actor_opt"]["state"].items():
        assert float(sd["state"][k]["step"]) == float(st["step"])
        assert torch.equal(sd["state"][k]["exp_avg"].cpu(), st["exp_avg"]) and torch.equal(sd["state"][k]["exp_avg_sq"].cpu(), st["exp_avg_sq"])
    torch.save(actor.state_dict(), tmp_path / "agent_0.th")
    back = torch.load(tmp_path / "agent_0.th", map_location="cpu")
    for k, v in g["actor"].items():
        assert torch.equal(back[k].cpu(), v), k


def test_reference_checkpoint_emulated(golden, tmp_path):
    check_reference_checkpoint(golden, "cpu", tmp_path)


def test_size_independent_properties_emulated():
    """tests/properties.py at small sizes through the host emulator (the GPU suite runs them at config 3's sizes)."""
    from iplan_amd.config import default_args
    from tests import properties as P
    args = default_args("highway", use_cuda=False, max_vehicle_num=3, n_agents=1, episode_limit=8, batch_size_run=2,
                        buffer_size=2, batch_size=1, max_history_len=3)
    loop = P.make_loop(args, 2, "cpu")
    batch = P.check_env_independence(loop, sub=1)
    P.check_behaviour_properties(loop, batch, fd_tol=5e-2)
    P.check_wgrad_additivity("cpu", n_nets=2, rows=40, n_inner=5, O=192, K=64)


@pytest.mark.parametrize("N", [2, 17])
def test_gat_entity_count_edges_emulated(N):
    from tests.test_gpu_gat import gat_vs_oracle
    gat_vs_oracle(B=2, N=N, D=13, seed=N, device="cpu")


@pytest.mark.parametrize("cfg", ["iplan", "gat_only", "plain", "tanh"])
def test_device_resident_rollout_matches_oracle_emulated(cfg):
    """The path bench.py times -- SyntheticLoop._rollout_body: in-place write_back / out= launches into the episode
    buffer -- against the oracle stepped the same way with the same draws, every field, every step (tests/rollout_oracle.py);
    config 3 (full iPLAN), config 2 (GAT on, Behaviour off) and config 1 (both off) feature layouts."""
    from iplan_amd.config import default_args
    from tests.rollout_oracle import check_rollout_body
    kw = dict(iplan={}, gat_only=dict(Behavior_enable=False), plain=dict(Behavior_enable=False, GAT_enable=False, GAT_use_behavior=False),
              tanh=dict(use_ReLU=False))[cfg]                  # (mlp.py:10: the actor/critic trunk's second activation)
    args = default_args("highway", use_cuda=False, max_vehicle_num=5, n_agents=2, episode_limit=4, batch_size_run=3,
                        max_history_len=3, **kw)
    worst = check_rollout_body(args, 3, "cpu", seed=5)
    assert max(worst.values()) < 1e-5


@pytest.mark.parametrize("kw", ["1", "2", "8"])
def test_device_resident_rollout_other_ksplit_wg_emulated(kw, monkeypatch):
    """The one-launch vector step with the action selection's K split over 1 / 2 / 8 workgroups per unit instead of the 4 the launch
    size picks (IplanAcFwdArgs.ksplit_wg; 1 = no cross-workgroup exchange: the shape of large env counts): every field against the
    oracle and against the two-launch form."""
    from iplan_amd.config import default_args
    from tests.rollout_oracle import check_rollout_body
    monkeypatch.setenv("IPLAN_AC_KSPLIT_WG", kw)
    args = default_args("highway", use_cuda=False, max_vehicle_num=5, n_agents=2, episode_limit=3, batch_size_run=3, max_history_len=3)
    worst = check_rollout_body(args, 3, "cpu", seed=7)
    assert max(worst.values()) < 1e-5


def test_fused_step_give_up_flag_raises_emulated():
    """sync[2] of a fused vector-step launch (a wait that hit its poll limit) is what the host refuses to go on with"""
    from iplan_amd import _lib as L, ops
    from iplan_amd.config import default_args
    from iplan_amd.harness import SyntheticLoop
    args = default_args("highway", use_cuda=False, max_vehicle_num=5, n_agents=2, episode_limit=3, batch_size_run=2, max_history_len=3)
    loop = SyntheticLoop(args, 2, seed=0, device="cpu")
    loop.rollout()
    assert ops._FUSED_SYNC and not ops.fused_sync_error()
    ops.check_fused_sync()
    buf = next(iter(ops._FUSED_SYNC.values()))
    buf[0], buf[2] = 1, 1                                  # (a producer that reported after the give-up: a count left behind)
    try:
        with pytest.raises(L.IplanError):
            ops.check_fused_sync()
        assert int(buf.abs().sum()) == 0                     # ... and the counters start clean again
        loop.rollout()
        ops.check_fused_sync()
    finally:
        buf.zero_()


def check_seq2seq(golden, device):
    """nova/Seq2Seq.py forward + backward (a19): same constructor / state_dict / random draws as the reference class, outputs and
    parameter gradients recorded from it"""
    import numpy as np
    from iplan_amd.nova.Seq2Seq import Seq2Seq
    for g in golden("seq2seq"):
        d = g["dims"]
        net = Seq2Seq(d["C"], d["H"], d["layers"], d["P"], num_node=5, output_size=d["O"], dropout=0.5, teacher_forcing_ratio=d["ratio"])
        assert list(net.state_dict().keys()) == list(g["params"].keys())
        net.load_state_dict(g["params"])
        np.random.seed(g["np_seed"])
        with torch.no_grad():
            out = net(g["x"].to(device), g["last"].to(device), g["teacher"].to(device), keep=g["masks"].reshape(d["P"], d["R"], d["H"]).to(device))
        assert out.shape == g["out"].shape and rel_err(out.cpu(), g["out"]) < 1e-5, (g["tag"], rel_err(out.cpu(), g["out"]))
        # training form: the saving forward gives the same prediction, and its backward the gradients the reference's autograd gave
        # under the recorded loss sum(out * gw) -- twice, to see .grad accumulate like torch's
        for rep in (1, 2):
            np.random.seed(g["np_seed"])
            out_t = net(g["x"].to(device), g["last"].to(device), g["teacher"].to(device), keep=g["masks"].reshape(d["P"], d["R"], d["H"]).to(device))
            assert out_t.requires_grad and torch.equal(out_t.detach().cpu(), out.cpu()), g["tag"]
            (out_t * g["gw"].to(device)).sum().backward()
            for k, p in net.named_parameters():
                e = rel_err(p.grad.cpu() / rep, g["grads"][k])
                assert e < 1e-5, (g["tag"], k, rep, e)
        # the reference's autograd would also differentiate w.r.t. the inputs; this path does not, and says so
        with pytest.raises(NotImplementedError, match="in_data.requires_grad"):
            net(g["x"].to(device).requires_grad_(True), g["last"].to(device), g["teacher"].to(device))
        with torch.no_grad():                                                              # (no graph wanted: fine)
            net(g["x"].to(device).requires_grad_(True), g["last"].to(device), g["teacher"].to(device))
        x4 = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)          # the reshape helpers (N, C, T, V)
        assert net.reshape_for_rnn(x4).shape == (10, 4, 3)
        assert torch.equal(net.reshape_from_rnn(net.reshape_for_rnn(x4)), x4)


def test_seq2seq_forward_and_backward_emulated(golden):
    check_seq2seq(golden, "cpu")


def check_seq2seq_shapes_vs_oracle(device, cases=None):
    """The shapes the fixtures do not hold -- the kernel's limits (4 layers, input width 64, output width 16), one-step encoder /
    decoder, evaluation mode (no dropout), no teacher, always-teacher -- against the oracle under fp64 autograd (output and every
    parameter gradient under a random linear loss)."""
    import numpy as np
    from oracle import iplan_oracle as O
    from iplan_amd.nova.Seq2Seq import Seq2Seq
    #         C   H  layers P  O   R  T  ratio train teacher
    cases = cases or [(64, 64, 4, 3, 16, 21, 2, 0.0, True, True), (3, 32, 3, 1, 1, 16, 1, 1.0, True, True), (5, 32, 2, 5, 2, 7, 3, 0.5, False, True),
                      (4, 64, 1, 4, 2, 33, 4, 0.5, True, False)]
    for i, (C, H, layers, P, No, R, T, ratio, train, with_teacher) in enumerate(cases):
        torch.manual_seed(100 + i)
        net = Seq2Seq(C, H, layers, P, num_node=1, output_size=No, dropout=0.25, teacher_forcing_ratio=ratio)
        net.train(train)
        gen = torch.Generator().manual_seed(200 + i)
        x = torch.rand(R, T, C, generator=gen) * 2 - 1
        last = torch.rand(R, 1, No, generator=gen) * 2 - 1
        teacher = torch.rand(R, P, No, generator=gen) * 2 - 1 if with_teacher else None
        gw = torch.rand(R, P, No, generator=gen) * 2 - 1
        keep = (torch.rand(P, R, H, generator=gen) > 0.25).float()
        p64 = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in net.state_dict().items()}
        np.random.seed(300 + i)
        coins = [bool(np.random.random() < ratio) for _ in range(P)]
        o64 = O.seq2seq_forward(p64, x.double(), last.double(), P, teacher.double() if with_teacher else None, coins,
                                keep.double().reshape(P, R, 1, H) if train else None, 0.25)
        (o64 * gw.double()).sum().backward()
        np.random.seed(300 + i)
        out = net(x.to(device), last.to(device), teacher.to(device) if with_teacher else None, keep=keep.to(device))
        tag = (C, H, layers, P, No, R, T, ratio, train, with_teacher)
        assert rel_err(out.detach().cpu(), o64.detach().float()) < 1e-5, (tag, rel_err(out.detach().cpu(), o64.detach().float()))
        (out * gw.to(device)).sum().backward()
        for k, q in net.named_parameters():
            e = rel_err(q.grad.cpu(), p64[k].grad.float())
            assert e < 1e-5, (tag, k, e)


def test_seq2seq_limit_shapes_vs_oracle_emulated():
    check_seq2seq_shapes_vs_oracle("cpu")


def test_fc1_pack_follows_parameter_writes_emulated(monkeypatch):
    """ops.Fc1Pack must repack after ANY write to fc1.weight / feature_norm.{weight, bias}: an in-place write through the
    Parameter (p.copy_ / p.mul_ under no_grad: what torch.optim or a re-initialisation does) is seen through the
    Parameter's own version counter; a write through a detached alias needs arena.touch()."""
    from iplan_amd import synth
    from iplan_amd.config import default_args
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    args = default_args("highway", use_cuda=False, max_vehicle_num=5, n_agents=2, episode_limit=3)
    torch.manual_seed(0)
    mac = DcntrlMAC(synth.make_scheme(args), {"agents": args.n_agents}, args)
    batch = synth.make_batch(args, 3, seed=1, device="cpu")

    def values(no_pack=False):
        if no_pack:
            monkeypatch.setenv("IPLAN_NO_FC1_PACK", "1")
        v = torch.as_tensor(mac.select_actions_ippo(batch, 1, test_mode=True)[0]).clone()
        monkeypatch.delenv("IPLAN_NO_FC1_PACK", raising=False)
        return v

    v0 = values()
    same = lambda a, b: (a - b).abs().max() < 2e-6 * max(1.0, float(b.abs().max()))   # noqa: E731  (packed vs in-place operands: summation order)
    assert same(v0, values(no_pack=True))
    w = dict(mac.critics[0].named_parameters())["base.mlp.fc1.0.weight"]
    with torch.no_grad():
        w.copy_(w + 0.05 * torch.randn_like(w))                     # through the Parameter: its _version moves (a uniform
                                                                    # rescaling would vanish in the LayerNorm behind fc1)
    v1 = values()
    assert (v1[:, 0] - v0[:, 0]).abs().max() > 1e-4 and same(v1, values(no_pack=True))
    g = dict(mac.critics[1].named_parameters())["base.feature_norm.weight"]
    g.data[:40].mul_(0.1)                                           # through a detached alias: invisible ...
    mac.critic_arena.touch()                                        # ... until the arena is told
    v2 = values()
    assert (v2[:, 1] - v1[:, 1]).abs().max() > 1e-4 and same(v2, values(no_pack=True))


def check_gumbel_noise(lib, device, n):
    """iplan_gumbel_noise: Gumbel(0, 1) moments (mean = Euler's gamma, variance = pi^2 / 6), no non-finite value, the same
    samples for the same seed whatever the launch covers (counter based), different ones for another seed"""
    import ctypes as C
    stream = L.current_stream(device)
    a, b, c = (torch.empty(n, dtype=torch.float32, device=device) for _ in range(3))
    for out, seed in ((a, 7), (b, 7), (c, 8)):
        assert lib.c.iplan_gumbel_noise(C.c_void_p(out.data_ptr()), C.c_int64(n), C.c_uint64(seed), C.c_void_p(stream)) == 0
    half = torch.empty(n // 2, dtype=torch.float32, device=device)
    assert lib.c.iplan_gumbel_noise(C.c_void_p(half.data_ptr()), C.c_int64(n // 2), C.c_uint64(7), C.c_void_p(stream)) == 0
    a, b, c, half = a.cpu().double(), b.cpu().double(), c.cpu().double(), half.cpu().double()
    assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a[:n // 2], half) and (a != c).float().mean() > 0.99
    tol = 6.0 * (1.6449 / n) ** 0.5
    assert abs(a.mean().item() - 0.5772157) < tol and abs(a.var().item() - 1.6449341) < 12 * tol, (a.mean().item(), a.var().item())
    # P(g <= x) = exp(-exp(-x))
    for x in (-1.0, 0.0, 1.0, 3.0):
        p = float(torch.exp(-torch.exp(torch.tensor(-x))))
        assert abs((a <= x).double().mean().item() - p) < 5.0 * (p * (1 - p) / n) ** 0.5 + 1e-9, x
    assert lib.c.iplan_gumbel_noise(C.c_void_p(a.data_ptr()), C.c_int64(3), C.c_uint64(1), C.c_void_p(stream)) != 0   # n % 4


def test_gumbel_noise_emulated():
    check_gumbel_noise(get_emu_lib(), "cpu", 1 << 15)


def check_fc1_split_vs_fp32(device, rows_per_ep=7, n_eps=5, seed=3, N=5, log=None):
    """The split-bf16 fc1 path of the PPO epochs (iplan_ac_xhat_pack / iplan_ac_fc1_split_fwd / iplan_ac_bwd_fc1_split)
    against the fp32 MFMA path of the same launch and against an fp64 evaluation of fc1(LayerNorm(x)): the forward's
    pre-activation and every fc1 / feature_norm gradient.  Ragged on purpose: rows not a multiple of 16 or 32, an odd number
    of k-tiles."""
    from iplan_amd import ops, synth
    from iplan_amd import _lib as L
    from iplan_amd.config import default_args
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    args = default_args("highway", use_cuda=(device != "cpu"), max_vehicle_num=N, n_agents=2, episode_limit=rows_per_ep)
    torch.manual_seed(seed)
    mac = DcntrlMAC(synth.make_scheme(args), {"agents": args.n_agents}, args)
    with torch.no_grad():                                         # LayerNorm(F) affine away from its (1, 0) initialisation
        for arena in (mac.actor_arena, mac.critic_arena):
            for m in arena.modules:
                p = dict(m.named_parameters())
                p["base.feature_norm.weight"].add_(0.3 * torch.randn_like(p["base.feature_norm.weight"]))
                p["base.feature_norm.bias"].add_(0.3 * torch.randn_like(p["base.feature_norm.bias"]))
            arena.touch()
    nA, T, T1, M = args.n_agents, rows_per_ep, rows_per_ep + 1, 64
    f = {k: v.to(device) for k, v in synth.make_episode_fields(args, n_eps, seed, 0.2).items()}
    srcs = []
    for key, w in mac._widths():
        t = f[key]
        srcs.append((t, w, t.stride(2), t.stride(1)))
    last = torch.randint(-1, args.n_actions, (n_eps, T1, nA), dtype=torch.int32, device=device)
    spec_all = ops.AcFeatureSpec(args.max_vehicle_num, srcs, n_actions=args.n_actions, last_action=last, la_strides=(1, nA),
                                 n_id=nA, T=T1, T_phys=T1)
    spec = ops.AcFeatureSpec(args.max_vehicle_num, srcs, n_actions=args.n_actions, last_action=last, la_strides=(1, nA),
                             n_id=nA, T=T, T_phys=T1)
    rows = n_eps * T
    h = torch.randn(2, n_eps, T1, nA, M, device=device) * 0.1
    hs = (h[0].stride(2), h[0].stride(1))
    ln_stats = torch.empty(nA, n_eps * T1, 2, device=device)
    ops.ac_forward(None, mac.critic_arena, 1, spec_all, n_eps * T1, nA, h_critic=h[1], h_strides=hs, ksplit=1, want_h=False,
                   ln_stats=ln_stats, ln_stats_mode=1)
    actions = torch.randint(0, args.n_actions, (n_eps, T1, nA, 1), device=device)
    kw = dict(h_actor=h[0], h_critic=h[1], h_strides=hs, mode=2, actions_in=actions, act_strides=(actions.stride(2), actions.stride(1)),
              n_actions=args.n_actions, ksplit=1, want_h=False, ln_stats=ln_stats, ln_stats_mode=2, save=True, want_entropy=True)
    xhat = ops.ac_xhat_pack(spec, rows, nA, ln_stats)
    g1, g2 = torch.randn(nA, rows, device=device), torch.randn(nA, rows, device=device)
    res = {}
    for tag, xh in (("fp32", None), ("split", xhat)):
        for arena in (mac.actor_arena, mac.critic_arena):
            arena.grad.zero_()
        out = ops.ac_forward(mac.actor_arena, mac.critic_arena, 2, spec, rows, nA, xhat=xh, **kw)
        bw = ops.ac_backward(out, mac.actor_arena, mac.critic_arena, g_logp=g1, g_entropy=-0.01 / rows, g_values=g2)
        res[tag] = dict(a1=out["saved"][..., 0:M].clone(), logp=out["logp"].clone(), values=out["values"].clone(),
                        dz1=bw["dsave"][..., 0:M].clone(), ga=mac.actor_arena.grad.clone(), gc=mac.critic_arena.grad.clone(),
                        z1=None if xh is None else out["_keep"][11].sum(0))          # ([K parts, 2, nA, rows, 64]: the parts add up)
    # fp64 evaluation from the raw fields
    x = torch.cat([torch.cat([f[key][:, :T].double() for key, _ in mac._widths()], dim=-1).flatten(-2)], dim=-1)   # [eps, T, nA, N*W]
    oh_last = torch.zeros(n_eps, T, nA, args.n_actions, dtype=torch.float64, device=device)
    lidx = last[:, :T].long()
    oh_last.scatter_(-1, lidx.clamp(min=0).unsqueeze(-1), (lidx >= 0).double().unsqueeze(-1))
    ident = torch.eye(nA, dtype=torch.float64, device=device).expand(n_eps, T, nA, nA)
    x = torch.cat([x, oh_last, ident], dim=-1)                     # [eps, T, nA, F]
    xhat64 = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    worst = {}
    for which, arena in enumerate((mac.actor_arena, mac.critic_arena)):
        for i in range(nA):
            W, gam, bet = (arena.param(i, n).double() for n in ("base.mlp.fc1.0.weight", "base.feature_norm.weight", "base.feature_norm.bias"))
            xr = xhat64[:, :, i].reshape(rows, -1)
            z64 = (xr * gam + bet) @ W.t()
            z = res["split"]["z1"][which, i].double()
            worst["z1"] = max(worst.get("z1", 0.0), rel_err(z, z64))
            b = arena.param(i, "base.mlp.fc1.0.bias").double()
            for tag in ("fp32", "split"):
                worst["a1_" + tag] = max(worst.get("a1_" + tag, 0.0), rel_err(res[tag]["a1"][which, i].double(), torch.relu(z64 + b)))
            # gradients of fc1.weight / feature_norm from each path's OWN dz1 (the tails differ by rounding), in fp64
            for tag in ("fp32", "split"):
                dz = res[tag]["dz1"][which, i].double()
                G = dz.t() @ xr
                S = dz.sum(0)
                ref = {"base.mlp.fc1.0.weight": gam * G + bet * S[:, None], "base.feature_norm.weight": (W * G).sum(0),
                       "base.feature_norm.bias": (W * S[:, None]).sum(0)}
                garena = res[tag]["ga" if which == 0 else "gc"]
                for name, r64 in ref.items():
                    off, numel = arena.off(name), r64.numel()
                    got = garena[i, off:off + numel].view(r64.shape).double()
                    worst[f"{name}_{tag}"] = max(worst.get(f"{name}_{tag}", 0.0), rel_err(got, r64))
    worst["logp_split_vs_fp32"] = rel_err(res["split"]["logp"], res["fp32"]["logp"])
    worst["values_split_vs_fp32"] = rel_err(res["split"]["values"], res["fp32"]["values"])
    if log is not None:
        log(worst)
    assert worst["logp_split_vs_fp32"] < 1e-5 and worst["values_split_vs_fp32"] < 1e-5, worst
    # every quantity within the parity contract's 1e-5 of the fp64 value, and the split path as close to fp64 as the fp32
    # MFMA path is (the K = F contraction is one fp32 chain there, 2.6e-6 of max|z| at F = 2485; six exact bf16 piece
    # products per K = 32 step accumulated in fp32 here)
    for k, v in worst.items():
        assert v < 1e-5, (k, v, worst)
    assert worst["z1"] <= max(2e-6, 1.5 * worst["a1_fp32"]), worst
    for name in ("a1", "base.mlp.fc1.0.weight", "base.feature_norm.weight", "base.feature_norm.bias"):
        assert worst[name + "_split"] <= max(2e-6, 1.5 * worst[name + "_fp32"]), (name, worst)
    return worst


@pytest.mark.parametrize("shape", ["full", "small"])
def test_fc1_split_vs_fp32_emulated(monkeypatch, shape):
    """both forward shapes at one size: four row tiles per wave (the full buffer's) and one (small batches: a data-parallel
    rank's share of the PPO rows)"""
    monkeypatch.setenv("IPLAN_AC_SPLIT_SHAPE", shape)
    check_fc1_split_vs_fp32("cpu")
    check_fc1_split_vs_fp32("cpu", rows_per_ep=13, n_eps=5, seed=4)     # 65 rows: three 32-row blocks, the last one ragged


def check_ac_ksplit_wg(device, N=5, E=19, monkeypatch=None):
    """The rollout-shaped actor / critic launch with its F-wide contraction split over 4 workgroups per (row tile, net) unit
    (IplanAcFwdArgs.ksplit_wg: partial sums through global memory, the last arrival runs the tail) against the one-workgroup
    form: same actions, values / log-probs to fp32 round-off; twice in a row (the ticket counters must be left at zero)."""
    from iplan_amd import synth
    from iplan_amd.config import default_args
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    args = default_args("highway", use_cuda=(device != "cpu"), max_vehicle_num=N, n_agents=3, episode_limit=4)
    torch.manual_seed(11)
    mac = DcntrlMAC(synth.make_scheme(args), {"agents": args.n_agents}, args)
    batch = synth.make_batch(args, E, seed=5, device=device)
    res = {}
    for kw in ("1", "4", "4", "2"):
        monkeypatch.setenv("IPLAN_AC_KSPLIT_WG", kw)
        out = mac.select_actions_ippo(batch, 1, test_mode=True, as_numpy=False)
        flat = []
        for o in out:
            flat += list(o) if isinstance(o, (list, tuple)) else [o]
        res.setdefault(kw, []).append([torch.as_tensor(o).clone().cpu() for o in flat])
    ref = res["1"][0]
    for kw, runs in res.items():
        for got in runs:
            for a, b in zip(got, ref):
                if a.dtype in (torch.int64, torch.int32):
                    assert torch.equal(a, b), kw
                else:
                    assert (a.double() - b.double()).abs().max() <= 2e-6 * max(1.0, float(b.abs().max())), (kw, float((a - b).abs().max()))
    for a, b in zip(res["4"][0], res["4"][1]):
        assert torch.equal(a, b)                                   # same launch twice: bit-identical


def test_ac_ksplit_wg_emulated(monkeypatch):
    check_ac_ksplit_wg("cpu", monkeypatch=monkeypatch)
    check_ac_ksplit_wg("cpu", N=9, E=35, monkeypatch=monkeypatch)     # three row tiles, the last one ragged
