"""Debug probe 3: where do post-train parameters / first-epoch gradients differ from the fp64 oracle (entry level)?"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from iplan_amd import ops, synth
from iplan_amd.config import default_args
from tests.oracle_checks import _fields, _Log, _sd, _req
from oracle import iplan_oracle as O

def run(epochs):
    args = default_args("highway", use_cuda=True, buffer_size=26, batch_size=25, ppo_epoch=epochs)
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    from iplan_amd.learners.ippo_learner import IPPOLearner
    torch.manual_seed(24)
    scheme = synth.make_scheme(args)
    mac = DcntrlMAC(scheme, {"agents": args.n_agents}, args)
    pre = dict(actors=[_sd(m) for m in mac.agents], critics=[_sd(m) for m in mac.critics])
    learner = IPPOLearner(mac, scheme, _Log(), args)
    E = args.buffer_size
    fields, batch = _fields(args, E, 25, 0.15, "cuda")
    learner.batch_size_run = E
    learner.insert_episode_batch(batch)
    learner.train(0)
    torch.cuda.synchronize()
    f64 = {k: (v.double() if v.is_floating_point() else v) for k, v in fields.items()}
    i = 0
    ap, cp = _req(pre["actors"][i], torch.float64), _req(pre["critics"][i], torch.float64)
    O.ppo_train_agent(i, ap, cp, f64, args)
    arena = mac.critic_arena
    for k in ("base.mlp.fc1.0.weight", "base.feature_norm.weight", "base.mlp.fc1.0.bias"):
        got_p = mac.critics[i].state_dict()[k].double().cpu()
        ref_p = cp[k].detach()
        got_g = arena.grad_of(i, k).double().cpu()
        ref_g = cp[k].grad
        dp, dg = (got_p - ref_p).abs(), (got_g - ref_g).abs()
        print(f"epochs={epochs} critic {k}: post max|d|={float(dp.max()):.3e} last-grad max|d|/max={float(dg.max() / ref_g.abs().max()):.3e}")
        flat = dp.flatten().topk(8).indices
        shp = got_p.shape
        for ix in flat.tolist():
            idx = (ix // shp[-1], ix % shp[-1]) if len(shp) == 2 else (ix,)
            print(f"    idx {idx}: p got {float(got_p[idx]):+.6e} ref {float(ref_p[idx]):+.6e} pre {float(pre['critics'][i][k][idx]):+.6e} | last g got {float(got_g[idx]):+.3e} ref {float(ref_g[idx]):+.3e}")

run(1)
run(2)
