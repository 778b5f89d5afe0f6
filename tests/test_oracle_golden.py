"""CPU: the oracle restatement reproduces the committed reference outputs (tests/golden/*.pt were
produced by oracle/make_golden.py running the real reference)."""
import torch

from oracle import iplan_oracle as O


def close(a, b, tol=1e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


def test_gat_forward_matches_reference(golden):
    for tag in ("small", "hwy", "wide"):
        g = golden("gat_" + tag)
        out = O.gat_forward(g["params"], g["obs"], g["h_prev"], g["noise"])
        assert close(out, g["out"]), tag


def test_gat_grads_match_reference(golden):
    g = golden("gat_small")
    p = {k: v.double().requires_grad_(True) for k, v in g["params"].items()}
    out = O.gat_forward(p, g["obs"].double(), g["h_prev"].double(), g["noise"].double())
    (out * g["gout"].double()).sum().backward()
    for k, ref in g["grads"].items():
        assert close(p[k].grad, ref, 2e-4), k


def test_encoder_decoder_match_reference(golden):
    g = golden("encoder")
    seq, hL, lat = O.encoder_forward(g["params"], g["x"], g["h0"])
    assert close(seq, g["seq"]) and close(hL, g["hL"]) and close(lat, g["latent"])
    g = golden("decoder")
    y, hT = O.decoder_forward(O.strip_prefix(g["params"], "decoder."), g["dec_in"], g["h0"], g["mask"], 0.1)
    assert close(y, g["y"]) and close(hT, g["hT"])
    g = golden("pred_decoder")
    pred = O.prediction_decoder_forward(g["params"], g["last"], g["hidden"], 5, g["masks"], 0.1)
    assert close(pred, g["pred"])


def test_behavior_learn_loss_matches_reference(golden):
    g = golden("behavior_learn")
    a = g["args"]
    hist = g["fields"]["history"][:, :-1]
    term = g["fields"]["terminated"][:, :-1]
    J = hist.shape[1] - 1 - a["max_history_len"]
    for i in range(a["n_agents"]):
        masks = torch.stack(g["dropout"][i * J:(i + 1) * J])
        beh, stab, _ = O.behavior_learn_loss(g["pre"]["enc"][i], g["pre"]["dec"][i], hist[:, :, i],
                                             term[:, :, i, 0], a["max_history_len"], a["soft_update_coef"],
                                             masks, a["decoder_dropout"])
        assert close(beh, g["behavior_loss"][i]) and close(stab, g["stability_loss"][i])


def test_behavior_learn_loss_env_shares_add_up(golden):
    """oracle.behavior_learn_loss(env_slice=...): the shares of a partition of the envs reproduce the reference-recorded loss
    values and the whole-batch fp64 gradient (the form the config-4 GPU test evaluates the oracle in), penalty on"""
    g = golden("behavior_learn")
    a = g["args"]
    hist = g["fields"]["history"][:, :-1].double()
    term = g["fields"]["terminated"][:, :-1]
    E, L = hist.shape[0], a["max_history_len"]
    J = hist.shape[1] - 1 - L
    assert E >= 2
    for i in range(a["n_agents"]):
        masks = torch.stack(g["dropout"][i * J:(i + 1) * J]).double()
        kw = (term[:, :, i, 0], L, a["soft_update_coef"], masks, a["decoder_dropout"], 0.3, 0.005)
        whole = [{k: v.double().clone().requires_grad_(True) for k, v in g["pre"][n][i].items()} for n in ("enc", "dec")]
        beh, stab, loss = O.behavior_learn_loss(whole[0], whole[1], hist[:, :, i], *kw)
        loss.backward()
        parts = [{k: v.double().clone().requires_grad_(True) for k, v in g["pre"][n][i].items()} for n in ("enc", "dec")]
        b_sum = s_sum = 0.0
        cut = (E + 1) // 2
        for sl in (slice(0, cut), slice(cut, E)):
            b_, s_, l_ = O.behavior_learn_loss(parts[0], parts[1], hist[:, :, i], *kw, env_slice=sl)
            l_.backward()
            b_sum, s_sum = b_sum + float(b_), s_sum + float(s_)
        assert close(b_sum, g["behavior_loss"][i]) and close(s_sum, g["stability_loss"][i])
        assert abs(b_sum - float(beh)) <= 1e-12 * max(1.0, abs(float(beh))) and abs(s_sum - float(stab)) <= 1e-12 * max(1.0, abs(float(stab)))
        for w, p in zip(whole, parts):
            for k in w:
                assert (w[k].grad - p[k].grad).abs().max().item() <= 1e-12 * max(1.0, w[k].grad.abs().max().item()), k


def test_huber_is_one_sided():
    e = torch.tensor([-20.0, -5.0, 5.0, 20.0])
    assert torch.allclose(O.huber_loss(e, 10.0), torch.tensor([0.0, 12.5, 12.5, 150.0]))


def test_behavior_hard_learn_loss_matches_reference(golden):
    g = golden("behavior_hard_learn")
    a = g["args"]
    hist = g["fields"]["history"][:, :-1]
    term = g["fields"]["terminated"][:, :-1]
    L = a["max_history_len"]
    J = hist.shape[1] // L - 1
    for i in range(a["n_agents"]):
        masks = torch.stack(g["dropout"][i * J:(i + 1) * J])
        loss = O.behavior_hard_learn_loss(g["pre"]["enc"][i], g["pre"]["dec"][i], hist[:, :, i], term[:, :, i, 0], L,
                                          masks, a["decoder_dropout"])
        assert close(loss, g["behavior_loss"][i])
