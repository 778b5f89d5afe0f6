"""Micro-benchmark: fused GAT forward at cfg3 (5 nets x 32 envs x 55 entities, D = 13)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iplan_amd import ops
from iplan_amd.arena import ParamArena
from iplan_amd.config import default_args
from iplan_amd.nova.GAT_Net import GAT_Net, gumbel_noise

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = default_args("highway", use_cuda=True)
nA, N, d, Z, A = 5, 55, 5, 8, 32
nets = [GAT_Net(d + Z, args) for _ in range(nA)]
arena = ParamArena(nets, "cuda")
hist = (torch.rand(E, nA, N, d, device="cuda") * 2 - 1).permute(1, 0, 2, 3)
lat = torch.softmax(torch.randn(E, nA, N, Z, device="cuda"), -1).permute(1, 0, 2, 3)
hid = (torch.randn(E, nA, N, A, device="cuda") * 0.1).permute(1, 0, 2, 3)
noise = gumbel_noise((nA, E, N, N - 1, 2), "cuda")
out = torch.empty(nA, E, N, A, device="cuda")
for _ in range(5):
    ops.gat_forward(arena, hist, lat, hid, noise, out=out)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 50
ev0.record()
for _ in range(iters):
    ops.gat_forward(arena, hist, lat, hid, noise, out=out)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / iters
V = E * N; P = E * N * (N - 1); H = 32; D = 13
flop = nA * (V * (2 * D * H + 24 * H * H + 6 * H * A + 12 * A * A) + P * (12 * H * H + 8 * H + 4 * A))
print(f"gat_fwd E={E}: {ms*1e3:.1f} us/launch, {flop/ms/1e9:.2f} TFLOP/s algorithmic ({flop/1e9:.2f} GFLOP)")
