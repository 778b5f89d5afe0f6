"""Size-independent properties of the hot path, used at small sizes through the host emulator (CPU suite) and at
BASELINE config 3's full sizes on the GPU (tests/test_gpu_fullsize.py) where no oracle run is affordable:

* env independence: no op on the path mixes envs, so the outputs for a subset of the envs (ragged: not a multiple of
  anything) are BITWISE those of the full launch (different grids, tiles and launch shapes);
* determinism: fixed-order reductions, no atomics -> a repeated learner step gives bitwise identical gradients;
* shard additivity (= the data-parallel contract): with the loss normalisers taken over the union (the `win_norm`
  override), the gradients of two env shards add up to the gradient of the full batch;
* directional derivative: loss(theta + eps g/|g|) - loss(theta - eps g/|g|) = 2 eps |g| for the analytic gradient g.
"""
import torch

from iplan_amd import ops
from iplan_amd.nova.GAT_Net import gumbel_noise


def rel(a, b):
    return (a - b).abs().max().item() / max(1e-12, b.abs().max().item())


def make_loop(args, E, device, seed=3):
    from iplan_amd.harness import SyntheticLoop
    return SyntheticLoop(args, E, seed=seed, device=device)


def check_env_independence(loop, sub):
    """GAT forward, rollout encoder and select_actions on envs [0, sub) vs the full E-env launch: bitwise equal."""
    a, E, dev = loop.args, loop.E, loop.device
    nA, N, Z, A = a.n_agents, a.max_vehicle_num, a.latent_dim, a.attention_dim
    g = torch.Generator(device="cpu").manual_seed(5)
    hist = loop.obs_sets[0]["hist"][0].permute(1, 0, 2, 3).contiguous()                  # [nA, E, N, d]
    lat = torch.softmax(torch.randn(nA, E, N, Z, generator=g), -1).to(dev)
    hid = (torch.randn(nA, E, N, A, generator=g) * 0.1).to(dev)
    noise = gumbel_noise((nA, E, N, N - 1, 2), "cpu").to(dev)
    full, _ = ops.gat_forward(loop.prediction.gat_arena, hist, lat, hid, noise)
    part, _ = ops.gat_forward(loop.prediction.gat_arena, hist[:, :sub].contiguous(), lat[:, :sub].contiguous(),
                              hid[:, :sub].contiguous(), noise[:, :sub].contiguous())
    assert torch.equal(full[:, :sub], part), "GAT forward depends on the other envs"
    window = loop.obs_sets[0]["hist"][0:a.max_history_len].permute(1, 2, 3, 0, 4).contiguous()   # [E, nA, N, L, d]
    eh = (torch.randn(E, 1, nA, N, a.encoder_rnn_dim, generator=g) * 0.1).to(dev)
    lat_e = lat.permute(1, 0, 2, 3).contiguous()
    l_full, h_full = loop.behavior.latent_update(window, eh, lat_e)
    l_part, h_part = loop.behavior.latent_update(window[:sub].contiguous(), eh[:sub].contiguous(), lat_e[:sub].contiguous())
    assert torch.equal(l_full[:sub], l_part) and torch.equal(h_full[:sub], h_part), "encoder depends on the other envs"
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        batch = loop.rollout()
    from iplan_amd import synth
    sub_batch = synth.DictBatch({k: v[:sub].contiguous() for k, v in batch.data.items()}, sub, batch.max_seq_length, dev)
    v1, a1, lp1, ha1, hc1 = loop.mac.select_actions_ippo(batch, 3, test_mode=True, as_numpy=False)
    v2, a2, lp2, ha2, hc2 = loop.mac.select_actions_ippo(sub_batch, 3, test_mode=True, as_numpy=False)
    t = torch.as_tensor
    assert torch.equal(t(v1)[:sub], t(v2)) and torch.equal(t(a1)[:sub], t(a2)), "actor/critic depend on the other envs"
    assert torch.equal(t(ha1)[:, :sub], t(ha2)) and torch.equal(t(hc1)[:, :sub], t(hc2))
    return batch


def _beh_inputs(loop, batch):
    a = loop.args
    history = batch["history"][:, :-1].to(dtype=torch.float32)
    term = batch["terminated"][:, :-1]
    mask = term[..., 0].permute(2, 0, 1).to(torch.float32).contiguous()                   # Highway polarity
    hist = history.permute(2, 0, 1, 3, 4)
    return hist, mask


def _beh_grads(loop, hist, mask, keep, win_norm=None):
    a = loop.args
    fwd = ops.beh_forward(loop.behavior.enc_arena, loop.behavior.dec_arena, hist, mask, a.max_history_len, a.latent_dim,
                          a.soft_update_coef, a.thres_small_variation, a.decoder_dropout, keep=keep, win_norm=win_norm)
    ops.beh_backward(loop.behavior.enc_arena, loop.behavior.dec_arena, fwd)
    return fwd["loss"].clone(), loop.behavior.enc_arena.grad.clone(), loop.behavior.dec_arena.grad.clone()


def check_behaviour_properties(loop, batch, fd_tol=2e-2):
    """determinism, shard additivity with global normalisers, directional derivative -- on Behavior_policy.learn's kernels"""
    a, E, dev = loop.args, loop.E, loop.device
    nA, N, Lw = a.n_agents, a.max_vehicle_num, a.max_history_len
    T = a.episode_limit
    J = T - 1 - Lw
    hist, mask = _beh_inputs(loop, batch)
    if float(mask.sum()) == 0:                                    # synthetic rollouts never terminate: give the loss a mask
        mask = (torch.rand(mask.shape, generator=torch.Generator().manual_seed(1)) < 0.5).float().to(dev)
    g = torch.Generator().manual_seed(9)
    keep = (torch.rand(nA, J, E * N, Lw, 64, generator=g) < 1.0 - a.decoder_dropout).to(torch.uint8).to(dev)
    loss, ge, gd = _beh_grads(loop, hist, mask, keep)
    loss2, ge2, gd2 = _beh_grads(loop, hist, mask, keep)
    assert torch.equal(ge, ge2) and torch.equal(gd, gd2) and torch.equal(loss, loss2), "behaviour learning is not deterministic"
    assert torch.isfinite(ge).all() and torch.isfinite(gd).all() and float(gd.abs().max()) > 0
    # two env shards with the union's normalisers add up to the full batch
    wn = ops.beh_window_mask_sums(mask, Lw)
    h = E // 2
    rows = lambda k, lo, hi: k[:, :, lo * N:hi * N].contiguous()   # noqa: E731
    _, ge_a, gd_a = _beh_grads(loop, hist[:, :h], mask[:, :h].contiguous(), rows(keep, 0, h), win_norm=wn)
    _, ge_b, gd_b = _beh_grads(loop, hist[:, h:], mask[:, h:].contiguous(), rows(keep, h, E), win_norm=wn)
    assert rel(ge_a + ge_b, ge) < 1e-5 and rel(gd_a + gd_b, gd) < 1e-5, (rel(ge_a + ge_b, ge), rel(gd_a + gd_b, gd))
    # directional derivative along the gradient, decoder parameters of every net at once
    arena = loop.behavior.dec_arena
    theta = arena.data.clone()
    gn = gd.reshape(nA, -1).norm(dim=1)                            # per net
    eps = 1e-2
    direction = gd / gn.reshape(nA, *([1] * (gd.dim() - 1)))
    def loss_at(sign):
        arena.data.copy_(theta + sign * eps * direction)
        f = ops.beh_forward(loop.behavior.enc_arena, arena, hist, mask, Lw, a.latent_dim, a.soft_update_coef,
                            a.thres_small_variation, a.decoder_dropout, keep=keep)
        return f["loss"][:, 0].clone()
    fd = (loss_at(+1) - loss_at(-1)) / (2 * eps)
    arena.data.copy_(theta)
    err = ((fd - gn).abs() / gn).max().item()
    assert err < fd_tol, ("directional derivative vs analytic gradient norm", fd.tolist(), gn.tolist())


def check_wgrad_additivity(device, n_nets=5, rows=4096, n_inner=16, O=192, K=64):
    """dW over all rows == dW(first half) accumulated with dW(second half) (beta = 1), incl. the recurrent x_shift."""
    g = torch.Generator().manual_seed(2)
    dy = torch.randn(n_nets, rows, n_inner, O, generator=g).to(device)
    x = torch.randn(n_nets, rows, n_inner, K, generator=g).to(device)
    P = O * K + O
    class A:                                                       # minimal arena stand-in: grad [n_nets, P]
        pass
    def run(ranges):
        grad = torch.zeros(n_nets, P, device=device)
        for k, (lo, hi) in enumerate(ranges):
            w = ops.Wgrad(grad, n_nets)
            st = (rows * n_inner * O, n_inner * O, O), (rows * n_inner * K, n_inner * K, K)
            w.add(dy.data_ptr() + 4 * lo * n_inner * O, st[0], O, hi - lo, n_inner, x=x.data_ptr() + 4 * lo * n_inner * K,
                  x_strides=st[1], K=K, x_shift=-1, dw_off=0, db_off=O * K, beta=0.0 if k == 0 else 1.0)
            w._keep += [dy, x]
            w.run()
        return grad
    full = run([(0, rows)])
    halves = run([(0, rows // 2), (rows // 2, rows)])
    assert rel(halves, full) < 1e-5, rel(halves, full)
    ref = torch.einsum("nrto,nrtk->nok", dy[:, :, 1:].double().cpu(), x[:, :, :-1].double().cpu()).reshape(n_nets, -1)
    assert rel(full[:, :O * K].double().cpu(), ref) < 1e-5
