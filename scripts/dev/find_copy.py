"""Which torch op is the 0.5 ms strided copy at the start of the learn phase?  (torch.profiler, one cycle)"""
import os, sys, contextlib, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from iplan_amd.config import default_args
from iplan_amd.harness import SyntheticLoop
args = default_args("highway", use_cuda=True)
loop = SyntheticLoop(args, 32, seed=0, device="cuda")
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(3):
        loop.cycle()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    with contextlib.redirect_stdout(io.StringIO()):
        loop.cycle()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_time_total > 100 and ("copy" in e.name or "to" == e.name or "contiguous" in e.name or "fill" in e.name or "zero" in e.name)]
for e in sorted(evs, key=lambda e: -e.device_time_total)[:12]:
    print(f"{e.name:28s} dev {e.device_time_total:8.1f} us  shapes {e.input_shapes}  stack {[s for s in (e.stack or []) if 'iplan_amd' in s][:3]}")
