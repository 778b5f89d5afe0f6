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
rows = []
for e in prof.events():
    if e.device_time_total > 150 and e.cpu_parent is not None:
        rows.append((e.device_time_total, e.name, e.input_shapes, [s for s in (e.stack or []) if "iplan_amd" in s][:2]))
for r in sorted(rows, key=lambda r: -r[0])[:25]:
    print(f"{r[0]:9.1f} us  {r[1]:32s} {r[2]}  {r[3]}")
