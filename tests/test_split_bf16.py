"""The arithmetic claim behind the split-bf16 contractions (iplan_amd/csrc/wave_tile.h: split_bf3 / mfma_bf16), checked on the
CPU with torch's round-to-nearest bfloat16 conversion -- the same rounding as v_cvt_pk_bf16_f32:

* an fp32 value is EXACTLY the sum of three bf16 pieces taken as round-to-nearest residuals;
* the six piece products a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0), each exact in fp32, reproduce an fp32 dot product
  to below fp32 round-off (the dropped terms are O(2^-26 |a||b|))."""
import torch


def split3(x):
    p0 = x.to(torch.bfloat16).float()
    r1 = x - p0
    p1 = r1.to(torch.bfloat16).float()
    p2 = (r1 - p1).to(torch.bfloat16).float()
    return p0, p1, p2


def test_three_bf16_pieces_are_exact():
    g = torch.Generator().manual_seed(0)
    for scale in (1e-20, 1e-6, 1e-2, 1.0, 37.5, 1e6, 1e20):
        x = (torch.rand(1 << 16, generator=g) * 2 - 1) * scale
        p0, p1, p2 = split3(x)
        assert torch.equal((p0.double() + p1.double()) + p2.double(), x.double()), scale
        assert torch.equal((p0 + p1) + p2, x), scale                      # also when re-added in fp32, largest first


def test_six_piece_products_match_fp32_contraction():
    g = torch.Generator().manual_seed(1)
    K = 32
    a = torch.randn(4096, K, generator=g)
    b = torch.randn(4096, K, generator=g) * 0.3
    a0, a1, a2 = split3(a)
    b0, b1, b2 = split3(b)
    exact = (a.double() * b.double()).sum(-1)
    # every piece product is exact in fp32 (8 x 8 significand bits)
    for x, y in ((a0, b0), (a0, b1), (a1, b0), (a1, b1), (a0, b2), (a2, b0)):
        assert torch.equal((x * y).double(), x.double() * y.double())
    six = sum((x.double() * y.double()).sum(-1) for x, y in ((a2, b0), (a0, b2), (a1, b1), (a1, b0), (a0, b1), (a0, b0)))
    scale = (a.abs().double() * b.abs().double()).sum(-1)
    dropped = ((six - exact).abs() / scale).max().item()
    assert dropped < 2.0 ** -25, dropped                                   # below an fp32 half-ulp (2^-24) of the products
    fp32_chain = torch.zeros(4096)
    for k in range(K):                                                      # what v_mfma_f32_16x16x4_f32 computes: an fp32 fma chain
        fp32_chain = torch.addcmul(fp32_chain, a[:, k], b[:, k])
    fp32_err = ((fp32_chain.double() - exact).abs() / scale).max().item()
    assert dropped < fp32_err, (dropped, fp32_err)
