"""CPU (host-emulated kernel build): the weight-gradient contraction against plain torch."""
import pytest
import torch

from iplan_amd import _lib as L
from iplan_amd import ops
from tests.emu.emu_lib import get_emu_lib


@pytest.fixture(autouse=True)
def emu():
    L.use_library_for_tests(get_emu_lib())
    yield
    L.use_library_for_tests(None)


def test_wgrad_dense_and_recurrent():
    torch.manual_seed(0)
    n_nets, n_outer, n_inner, H = 2, 7, 5, 32
    dy = torch.randn(n_nets, n_outer, n_inner, 4 * H)          # [dr, dz, dn_i, dn_h]
    hs = torch.randn(n_nets, n_outer, n_inner, H)
    h0 = torch.randn(n_nets, n_outer, H)
    xin = torch.randn(n_nets, n_outer, n_inner, 13)
    grad = torch.zeros(n_nets, 20000)
    grad[:, 5000:5000 + 96 * 13] = 1.0                           # beta = 1 target
    w = ops.Wgrad(grad, n_nets)
    st = (dy.stride(0), dy.stride(1), dy.stride(2))
    # W_hh: dgh = [dr, dz, dn_h] against h_{t-1} (shift -1, h0 for the first step); bias too
    w.add(dy, st, 3 * H, n_outer, n_inner, x=hs, x_strides=(hs.stride(0), hs.stride(1), hs.stride(2)), K=H,
          dw_off=0, db_off=4000, seg=(2 * H, 0, 3 * H), x_shift=-1, x0=h0, x0_strides=(h0.stride(0), h0.stride(1)))
    # W_ih: dgi = [dr, dz, dn_i] against the step input (odd K), accumulated, scaled
    w.add(dy, st, 3 * H, n_outer, n_inner, x=xin, x_strides=(xin.stride(0), xin.stride(1), xin.stride(2)), K=13,
          dw_off=5000, db_off=-1, beta=1.0, scale=0.5)
    # reverse-direction recurrent weight: previous state is step + 1, zeros beyond the end; strided dW
    w.add(dy, st, 3 * H, n_outer, n_inner, x=hs, x_strides=(hs.stride(0), hs.stride(1), hs.stride(2)), K=H,
          dw_off=8000, dw_ld=2 * H, dw_col0=H, x_shift=1)
    # bias only
    w.add(dy, st, 4 * H, n_outer, n_inner, db_off=16000)
    w.run()
    for n in range(n_nets):
        d = dy[n].reshape(-1, 4 * H)
        dgh = torch.cat([d[:, :2 * H], d[:, 3 * H:]], 1)
        hprev = torch.cat([h0[n][:, None], hs[n][:, :-1]], 1).reshape(-1, H)
        ref = dgh.t() @ hprev
        assert torch.allclose(grad[n, :96 * H].view(96, H), ref, rtol=1e-5, atol=1e-5)
        assert torch.allclose(grad[n, 4000:4096], dgh.sum(0), rtol=1e-5, atol=1e-5)
        ref = 1.0 + 0.5 * (d[:, :3 * H].t() @ xin[n].reshape(-1, 13))
        assert torch.allclose(grad[n, 5000:5000 + 96 * 13].view(96, 13), ref, rtol=1e-5, atol=1e-5)
        hnext = torch.cat([hs[n][:, 1:], torch.zeros(n_outer, 1, H)], 1).reshape(-1, H)
        ref = d[:, :3 * H].t() @ hnext
        got = grad[n, 8000:8000 + 96 * 2 * H].view(96, 2 * H)
        assert torch.allclose(got[:, H:], ref, rtol=1e-5, atol=1e-5)
        assert got[:, :H].abs().max() == 0
        assert torch.allclose(grad[n, 16000:16000 + 4 * H], d.sum(0), rtol=1e-5, atol=1e-5)


def test_wgrad_many_rows_chunked():
    torch.manual_seed(1)
    R = 1000
    dy = torch.randn(1, R, 1, 20)
    x = torch.randn(1, R, 1, 70)
    grad = torch.zeros(1, 4000)
    ops.Wgrad(grad, 1).add(dy, (0, 20, 20), 20, R, 1, x=x, x_strides=(0, 70, 70), K=70, dw_off=0, db_off=3000).run()
    ref = dy[0, :, 0].t() @ x[0, :, 0]
    assert torch.allclose(grad[0, :1400].view(20, 70), ref, rtol=1e-5, atol=1e-4)
    assert torch.allclose(grad[0, 3000:3020], dy[0, :, 0].sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("O,K,rows,n_inner,shift", [
    (5, 64, 37, 3, 0),         # thin along O  (1 x 4 tiles), ragged rows
    (64, 13, 1000, 1, 0),      # thin along K  (4 x 1), many 8x-short chunks
    (192, 64, 300, 5, -1),     # wide (12 x 4), recurrent operand
    (130, 40, 45, 9, 1),       # wide with ragged tiles on both axes, reverse shift
    (100, 70, 333, 1, 0),      # square jobs (7 x 5 tiles -> 2 x 2 jobs)
    (16, 16, 16, 1, 0),        # exactly one tile, one block
    (7, 0, 50, 2, 0),          # bias only
])
def test_wgrad_job_shapes_vs_torch(O, K, rows, n_inner, shift):
    """every job shape of wgrad.hip (wide / square / thin), ragged edges, row tails and the shifted recurrent operand"""
    torch.manual_seed(O * 1000 + K)
    n_nets = 2
    dy = torch.randn(n_nets, rows, n_inner, O + 3)               # row pitch wider than O: extra columns must be ignored
    x = torch.randn(n_nets, rows, n_inner, K + 5) if K else None
    grad = torch.zeros(n_nets, O * (K + 1) + 8)
    w = ops.Wgrad(grad, n_nets)
    w.add(dy, (dy.stride(0), dy.stride(1), dy.stride(2)), O, rows, n_inner, x=x,
          x_strides=(x.stride(0), x.stride(1), x.stride(2)) if K else (0, 0, 0), K=K, x_col0=2 if K else 0, x_shift=shift,
          dw_off=0 if K else -1, db_off=O * K)
    w.run()
    for n in range(n_nets):
        d = dy[n, :, :, :O].double()
        assert torch.allclose(grad[n, O * K:O * K + O].double(), d.reshape(-1, O).sum(0), rtol=1e-5, atol=1e-4)
        if K:
            xs = x[n, :, :, 2:2 + K].double()
            if shift == -1:
                xs = torch.cat([torch.zeros(rows, 1, K, dtype=torch.double), xs[:, :-1]], 1)
            elif shift == 1:
                xs = torch.cat([xs[:, 1:], torch.zeros(rows, 1, K, dtype=torch.double)], 1)
            ref = torch.einsum("rto,rtk->ok", d, xs)
            assert torch.allclose(grad[n, :O * K].view(O, K).double(), ref, rtol=1e-5, atol=2e-4), (O, K)


@pytest.mark.parametrize("O,K,s0", [(192, 64, 0), (192, 64, 3), (40, 13, 0), (5, 64, 2)])
def test_wgrad_column_grouped_operands(O, K, s0):
    """column-grouped operands (the behaviour decoder's records: [tile][16-column group][step][chain][16]; rows = (tile,
    step * 16 + chain)), dY columns through the segment map, X through x_col0, and the recurrent operand read 16 rows back --
    in place where the rows in front of a window range exist (x_pre_valid), zeros in front of step 0"""
    torch.manual_seed(O + K + s0)
    n_nets, tiles, steps, Gd, Gx = 2, 3, 7, 14, 9                 # 224 dY columns, 144 X columns
    dy = torch.randn(n_nets, tiles, Gd, steps, 16, 16)
    x = torch.randn(n_nets, tiles, Gx, steps, 16, 16)
    c0, xc0 = 16, 32                                                # first dY column / first X column of the problem
    n = steps - s0
    grad = torch.zeros(n_nets, O * (K + 1) + 8)
    w = ops.Wgrad(grad, n_nets)
    st = lambda t, G: (tiles * G * steps * 256, G * steps * 256, 16)   # noqa: E731
    w.add(dy.data_ptr() + 4 * s0 * 256, st(dy, Gd), O, tiles, n * 16, x=x.data_ptr() + 4 * s0 * 256, x_strides=st(x, Gx), K=K, x_col0=xc0,
          x_shift=-16, x_pre_valid=s0 > 0, seg=(O, c0, 0), dw_off=0, db_off=O * K, dy_cg_stride=steps * 256, x_cg_stride=steps * 256)
    w._keep += [dy, x]
    w.run()
    rows = lambda t: t.permute(0, 1, 3, 4, 2, 5).reshape(n_nets, tiles, steps, 16, -1).double()   # noqa: E731  [net, tile, step, chain, col]
    d, xs = rows(dy)[..., c0:c0 + O], rows(x)[..., xc0:xc0 + K]
    xprev = torch.cat([torch.zeros_like(xs[:, :, :1]), xs[:, :, :-1]], 2)                          # previous step of the same chain
    for k in range(n_nets):
        ref = torch.einsum("tsco,tsck->ok", d[k, :, s0:], xprev[k, :, s0:])
        assert torch.allclose(grad[k, :O * K].view(O, K).double(), ref, rtol=1e-5, atol=2e-4), (O, K, s0)
        assert torch.allclose(grad[k, O * K:O * K + O].double(), d[k, :, s0:].sum((0, 1, 2)), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("s0,steps,tiles", [(0, 7, 3), (3, 9, 2), (0, 70, 1)])
def test_wgrad_gru_pair_shares_the_gate_gradients(s0, steps, tiles, monkeypatch):
    """the two wide problems of a 64-wide GRU on column-grouped records (the behaviour decoder's deferred update: dW_ih from
    [dr dz dn_i] x u, dW_hh from [dr dz dn_h] x h_{t-1}) run as ONE paired launch that fetches [dr dz] once
    (wgrad_pair_bf16_kernel): against torch, and BIT-identical to the two unpaired jobs (IPLAN_WG_NO_PAIR=1); ragged row tail
    (rows not a multiple of 32), a window range that starts past step 0, several row chunks"""
    torch.manual_seed(steps * 10 + s0)
    n_nets, H, Gd, Gx = 2, 64, 21, 31
    dy = torch.randn(n_nets, tiles, Gd, steps, 16, 16)              # 336 dY columns: [.. 80 | dr dz dn_i (192) | dn_h (64)]
    x = torch.randn(n_nets, tiles, Gx, steps, 16, 16)               # 496 record columns: u at 32, h at 352
    n = steps - s0
    st = lambda G: (tiles * G * steps * 256, G * steps * 256, 16)   # noqa: E731
    cg = dict(dy_cg_stride=steps * 256, x_cg_stride=steps * 256)

    def run():
        grad = torch.zeros(n_nets, 2 * (3 * H * (H + 1)) + 8)
        w = ops.Wgrad(grad, n_nets)
        ddp, sdp = dy.data_ptr() + 4 * s0 * 256, x.data_ptr() + 4 * s0 * 256
        w.add(ddp, st(Gd), 3 * H, tiles, n * 16, x=sdp, x_strides=st(Gx), K=H, x_col0=32, seg=(3 * H, 80, 0), dw_off=0, db_off=3 * H * H, **cg)
        w.add(ddp, st(Gd), 3 * H, tiles, n * 16, x=sdp, x_strides=st(Gx), K=H, x_col0=352, x_shift=-16, x_pre_valid=s0 > 0,
              seg=(2 * H, 80, 80 + 3 * H), dw_off=3 * H * (H + 1), db_off=3 * H * (H + 1) + 3 * H * H, **cg)
        w._keep += [dy, x]
        w.run()
        return grad

    paired = run()
    monkeypatch.setenv("IPLAN_WG_NO_PAIR", "1")
    plain = run()
    assert torch.equal(paired, plain)
    rows = lambda t: t.permute(0, 1, 3, 4, 2, 5).reshape(n_nets, tiles, steps, 16, -1).double()   # noqa: E731
    d, xs = rows(dy), rows(x)
    u, h = xs[..., 32:32 + H], xs[..., 352:352 + H]
    hprev = torch.cat([torch.zeros_like(h[:, :, :1]), h[:, :, :-1]], 2)
    dgi, dgh = d[..., 80:80 + 3 * H], torch.cat([d[..., 80:80 + 2 * H], d[..., 272:272 + H]], -1)
    o2 = 3 * H * (H + 1)
    for k in range(n_nets):
        for off, dg, xx in ((0, dgi, u), (o2, dgh, hprev)):
            ref = torch.einsum("tsco,tsck->ok", dg[k, :, s0:], xx[k, :, s0:])
            assert torch.allclose(paired[k, off:off + 3 * H * H].view(3 * H, H).double(), ref, rtol=1e-5, atol=5e-4)
            assert torch.allclose(paired[k, off + 3 * H * H:off + 3 * H * (H + 1)].double(), dg[k, :, s0:].sum((0, 1, 2)), rtol=1e-5, atol=2e-4)
