"""debug: Seq2Seq gradients vs the oracle per parameter for a few shapes (GPU)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import iplan_oracle as O
from iplan_amd.nova.Seq2Seq import Seq2Seq
dev = sys.argv[1] if len(sys.argv) > 1 else "cuda"
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for (C, H, layers, P, No, R, T) in [(64, 64, 4, 3, 16, 21, 2), (64, 64, 1, 3, 16, 21, 2), (4, 64, 1, 3, 16, 21, 2), (64, 64, 1, 3, 2, 21, 2), (64, 64, 1, 3, 2, 32, 2), (64, 64, 2, 3, 2, 21, 2), (64, 32, 4, 3, 16, 21, 2)]:
    torch.manual_seed(1)
    net = Seq2Seq(C, H, layers, P, num_node=1, output_size=No, dropout=0.25, teacher_forcing_ratio=0.0)
    net.eval()
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(R, T, C, generator=gen) * 2 - 1
    last = torch.rand(R, 1, No, generator=gen) * 2 - 1
    gw = torch.rand(R, P, No, generator=gen) * 2 - 1
    p64 = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    o64 = O.seq2seq_forward(p64, x.double(), last.double(), P, None, None, None, 0.0)
    (o64 * gw.double()).sum().backward()
    out = net(x.to(dev), last.to(dev))
    (out * gw.to(dev)).sum().backward()
    print((C, H, layers, P, No, R, T), "out", "%.1e" % rel(out.detach().cpu(), o64.detach().float()),
          " ".join("%s=%.1e" % (k.replace("coder.rnn.", "").replace("weight_", "W").replace("bias_", "b"), rel(q.grad.cpu(), p64[k].grad.float())) for k, q in net.named_parameters()))
