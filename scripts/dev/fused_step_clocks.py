"""Diagnostic: where the time of one fused vector step goes (wall_clock64 stamps, 100 MHz): the GAT scenes' phases and, per actor/critic
workgroup, entry / wait begin / wait end / contraction done / partial sums exchanged / tail done, all relative to the launch's first stamp."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from iplan_amd import ops  # noqa: E402
from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.harness import SyntheticLoop  # noqa: E402

args = default_args("highway", use_cuda=True, batch_size_run=32)
E, dev = 32, "cuda"
loop = SyntheticLoop(args, E, seed=0, device=dev)
batch = loop.rollout()
torch.cuda.synchronize()
D, a = batch.data, args
nA, N, L = a.n_agents, a.max_vehicle_num, a.max_history_len
hist_all = loop.obs_sets[0]["hist"]
eh = torch.zeros(2, E, 1, nA, N, a.encoder_rnn_dim, device=dev)
from iplan_amd.nova.GAT_Net import gumbel_noise  # noqa: E402
noise = gumbel_noise((nA, E, N, N - 1, 2), dev)
q = torch.empty(nA, E, a.n_actions, device=dev).exponential_()
t = 5
for rep in range(3):
    gclk = torch.zeros(nA * E * 5, dtype=torch.int64, device=dev)
    aclk = torch.zeros(4096, dtype=torch.int64, device=dev)
    window = hist_all[t + 1:t + 1 + L].permute(1, 2, 3, 0, 4)
    enc = loop.behavior.latent_update(window, eh[t & 1], D["behavior_latent"][:, t], out_latent=D["behavior_latent"][:, t + 1],
                                      out_hidden=eh[(t + 1) & 1][:, 0], launch=False)
    nxt = loop.mac.select_actions_ippo(batch, t + 1, test_mode=False, q_noise=q, as_numpy=False, write_back=True, launch=False, phase_clocks=aclk)
    hist = D["history"][:, t + 1].permute(1, 0, 2, 3)
    lat = D["behavior_latent"][:, t].permute(1, 0, 2, 3)
    hid = D["attention_latent"][:, t].permute(1, 0, 2, 3)
    ops.gat_forward(loop.prediction.gat_arena, hist, lat, hid, noise, out=D["attention_latent"][:, t + 1].permute(1, 0, 2, 3),
                    phase_clocks=gclk, fuse_enc=enc, fuse_ac=nxt)
    torch.cuda.synchronize()
g = gclk.view(-1, 5).cpu().double()
n_ac = int((aclk.view(-1, 8)[:, 0] != 0).sum())
c = aclk.view(-1, 8)[:n_ac].cpu().double()
t0 = min(g[:, 0].min().item(), c[:, 0].min().item())
us = lambda x: (x - t0) / 100.0
print(f"GAT scenes: start {us(g[:, 0]).min():.1f}..{us(g[:, 0]).max():.1f} us, end {us(g[:, 4]).min():.1f}..{us(g[:, 4]).max():.1f} us; "
      f"phases (mean us): " + " ".join(f"{(g[:, i + 1] - g[:, i]).mean().item() / 100:.1f}" for i in range(4)))
print(f"{n_ac} actor/critic workgroups")
names = ["entry", "wait begin", "wait end", "contraction done", "exchanged (last arrival)", "tail done"]
for i, nm in enumerate(names):
    col = c[:, i]
    col = col[col != 0]
    if len(col):
        print(f"  {nm:26s} n={len(col):3d}  min {us(col).min():7.1f}  mean {us(col).mean():7.1f}  max {us(col).max():7.1f} us")
last = c[c[:, 5] != 0]
print("  last arrivals: wait end -> contraction", ((last[:, 3] - last[:, 2]) / 100).mean().item(), "us; -> exchanged", ((last[:, 4] - last[:, 3]) / 100).mean().item(),
      "us; tail", ((last[:, 5] - last[:, 4]) / 100).mean().item(), "us")
print("sync error", ops.fused_sync_error())
