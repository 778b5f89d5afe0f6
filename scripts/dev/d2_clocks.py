"""Diagnostic: segment clocks of beh_dec_fwd2_kernel (library built with -DD2_CLOCKS, IPLAN_HIP_LIB=build/abl/lib_1.so)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
os.environ["IPLAN_BEH_SERIAL"] = "1"
from iplan_amd import _lib as L
from iplan_amd.config import default_args
from iplan_amd.harness import SyntheticLoop
args = default_args("highway", use_cuda=True)
loop = SyntheticLoop(args, 32, seed=0, device="cuda")
batch = loop.rollout()
torch.cuda.synchronize()
for it in range(2):
    loop.behavior.learn(batch, 0)
    loop.behavior.join_decoder()
    torch.cuda.synchronize()
out = (C.c_longlong * 16)()
rc = L.get_lib().c.iplan_debug_d2_clocks(out)
v = list(out)
J, Lw = args.episode_limit - 1 - args.max_history_len, args.max_history_len
steps = J * Lw / 4          # serial mode may still run in 4 window-range launches; clocks are those of the LAST launch
namesB = ["wait h, wait gi", "read h / gi + 108 MFMA", "signal + gates + split + publish h", "stores + keep + act + y MFMA", "wait y slots", "publish y", "-", "loop / address"]
namesA = ["masks + lin + stores + publish u", "wait u", "read u + 108 MFMA", "wait gi slot", "write gi + signal", "finish_y", "-", "loop / address"]
print("rc", rc, "B total", sum(v[:8]), "A total", sum(v[8:]))
for nm, x in zip(namesB, v[:8]): print(f"  B {nm:34s} {x:12d} cycles  {100.0 * x / max(1, sum(v[:8])):5.1f} %")
for nm, x in zip(namesA, v[8:]): print(f"  A {nm:34s} {x:12d} cycles  {100.0 * x / max(1, sum(v[8:])):5.1f} %")
