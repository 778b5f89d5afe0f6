import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler; faulthandler.dump_traceback_later(50, repeat=True)
import torch
from iplan_amd.config import default_args
from iplan_amd.harness import SyntheticLoop
t=time.time()
args = default_args("highway", use_cuda=True, batch_size_run=32, episode_limit=int(sys.argv[1]) if len(sys.argv)>1 else 6)
loop = SyntheticLoop(args, 32, seed=0, device="cuda")
torch.cuda.synchronize(); print("construct", time.time()-t, flush=True)
for it in range(3):
    t=time.time(); b = loop.rollout(); torch.cuda.synchronize(); print("rollout", it, time.time()-t, flush=True)
# per-call timings
import numpy as np
E=32; a=args; dev="cuda"
att = torch.zeros(E, 5, 55, 32, device=dev); lat = torch.zeros(E, 5, 55, 8, device=dev); eh = torch.zeros(E,1,5,55,32, device=dev)
single = loop.obs_sets[0]["hist"][9]
window = loop.obs_sets[0]["hist"][0:10].permute(1,2,3,0,4).contiguous()
def tm(name, fn, n=20):
    fn(); torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.time()-t)/n*1e3:.3f} ms", flush=True)
tm("GAT_latent_update", lambda: loop.prediction.GAT_latent_update(single, att, lat))
tm("latent_update", lambda: loop.behavior.latent_update(window, eh, lat))
tm("select_actions", lambda: loop.mac.select_actions_ippo(b, 1, test_mode=False, as_numpy=False))
