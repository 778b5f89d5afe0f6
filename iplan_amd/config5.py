"""BASELINE.json config 5: "Synthetic 64 agents x 63 neighbours x obs_dim 128, GAT+behavior forward/backward only, rocprof
roofline run" (B in {32, 256} env rows, 5 nets, N = 64, D = 128, H = A = 32, Z = 8, L = 10) as a function:
``measure()`` times, with HIP events on the launch stream, the GAT forward (rollout form), the GAT forward + backward (training
form, incl. its weight-gradient contraction) and the behaviour encoder / decoder forward + BPTT (Behavior_policy's kernels; full
90-step episode at B = 32, a 30-step episode at B = 256 to bound the activation records) and returns algorithmic FLOPs / achieved
TFLOP/s / fraction of the fp32-MFMA peak per piece (SURVEY.md section 8d FLOP convention).  Used by scripts/cfg5_bench.py (the
rocprofv3 run) and by bench.py, which appends the rows to its JSON line as ``config5`` (outside the timed region)."""
import torch

from . import ops
from .arena import ParamArena
from .config import default_args
from .nova.GAT_Net import GAT_Net, gumbel_noise
from .nova.behavior_net import Behavior_Latent_Decoder, EncoderRNN

PEAK = 157.3  # TFLOP/s fp32 MFMA (MI355X_MICROARCH.md)


def timed(fn, n, warm=1):
    """mean GPU time of fn() over n calls, each bracketed by its own events with a device synchronise in between: the BPTT
    records of one call at these sizes are tens of GB that the caching allocator can only hand back once the side streams that
    touched them are idle -- without the synchronise the next call pays hipMalloc for all of it (a benchmark artefact: the
    training loop re-uses the blocks)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e-3


def gat_flops(n_nets, B, N, D, H=32, A=32):
    V, P = B * N, B * N * (N - 1)
    return n_nets * (V * (2 * D * H + 24 * H * H + 6 * H * A + 12 * A * A) + P * (12 * H * H + 8 * H + 4 * A))


def measure(Bs=(32, 256), pieces=("gat", "beh"), reps=1.0, device="cuda"):
    """-> list of dict(piece, B, ms, gflop, tflops, frac).  ``reps`` scales the repetition counts (bench.py uses fewer)."""
    dev = torch.device(device)
    nN, N, D, d, Z, A, L = 5, 64, 128, 5, 8, 32, 10
    args = default_args("highway", use_cuda=True, max_vehicle_num=N, n_agents=nN)
    torch.manual_seed(0)
    rows = []
    rn = lambda n: max(1, int(round(n * reps)))  # noqa: E731
    for B in Bs:
        if "gat" in pieces:
            arena = ParamArena([GAT_Net(D, args) for _ in range(nN)], dev)
            obs = torch.rand(nN, B, N, D, device=dev) * 2 - 1
            hid = torch.randn(nN, B, N, A, device=dev) * 0.1
            noise = gumbel_noise((nN, B, N, N - 1, 2), dev)
            out = torch.empty(nN, B, N, A, device=dev)
            gout = torch.randn(nN, B, N, A, device=dev)
            f = gat_flops(nN, B, N, D)
            t = timed(lambda: ops.gat_forward(arena, obs, None, hid, noise, out=out), rn(20), 3)
            rows.append(dict(piece="gat_fwd", B=B, ms=t * 1e3, gflop=f / 1e9, tflops=f / t / 1e12, frac=f / t / 1e12 / PEAK))

            def fb():
                o, saved = ops.gat_forward(arena, obs, None, hid, noise, save=True)
                ops.gat_backward(arena, saved, gout)
            t = timed(fb, rn(10), 2)                      # (4 ms per call: more of them cost nothing; with 3 the mean moved by 6 % between runs)
            rows.append(dict(piece="gat_fwd+bwd(+wgrad)", B=B, ms=t * 1e3, gflop=3 * f / 1e9, tflops=3 * f / t / 1e12,
                             frac=3 * f / t / 1e12 / PEAK))
            del obs, hid, noise, out, gout
        if "beh" in pieces:
            T = 90 if B <= 32 else 30
            J = T - 1 - L
            enc = ParamArena([EncoderRNN(d, 32, Z, 1) for _ in range(nN)], dev)
            dec = ParamArena([Behavior_Latent_Decoder(d + Z, 64, 1, d, 0.1) for _ in range(nN)], dev)
            hist = torch.rand(nN, B, T, N, d, device=dev) * 2 - 1
            mask = (torch.rand(nN, B, T, device=dev) < 0.5).float()
            V = B * N
            f_dec = nN * V * J * L * (2 * (d + Z) * 64 + 12 * 64 * 64 + 2 * 64 * d)
            f_enc = nN * V * J * (L * (2 * d * 32 + 12 * 32 * 32) + 2 * 32 * Z)
            t = timed(lambda: ops.beh_forward(enc, dec, hist, mask, L, Z, 0.1, 0.005, 0.1, seed=1), rn(3), 1)
            rows.append(dict(piece=f"beh_fwd(T={T})", B=B, ms=t * 1e3, gflop=(f_dec + f_enc) / 1e9, tflops=(f_dec + f_enc) / t / 1e12,
                             frac=(f_dec + f_enc) / t / 1e12 / PEAK))

            def fb():
                fw = ops.beh_forward(enc, dec, hist, mask, L, Z, 0.1, 0.005, 0.1, seed=1)
                ops.beh_backward(enc, dec, fw)
            t = timed(fb, rn(3), 1)
            ff = 3 * (f_dec + f_enc)
            rows.append(dict(piece=f"beh_fwd+bwd(+wgrad)(T={T})", B=B, ms=t * 1e3, gflop=ff / 1e9, tflops=ff / t / 1e12, frac=ff / t / 1e12 / PEAK))
            del hist, mask
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return rows
