"""Debug probe 4: same flow as tests.oracle_checks.check_ppo_train_vs_oracle (no heavy spy): is the final gradient arena what
ac_backward produced?  is train() deterministic run to run?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from iplan_amd import ops, synth
from iplan_amd.config import default_args
from tests.oracle_checks import _fields, _Log

def run(light_spy):
    args = default_args("highway", use_cuda=True, ppo_epoch=2)
    from iplan_amd.controllers.dcntrl_controller import DcntrlMAC
    from iplan_amd.learners.ippo_learner import IPPOLearner
    torch.manual_seed(24)
    scheme = synth.make_scheme(args)
    mac = DcntrlMAC(scheme, {"agents": args.n_agents}, args)
    learner = IPPOLearner(mac, scheme, _Log(), args)
    E = args.buffer_size
    fields, batch = _fields(args, E, 25, 0.15, "cuda")
    learner.batch_size_run = E
    learner.insert_episode_batch(batch)
    rec = []
    if light_spy:
        orig = ops.ac_backward
        def spy(fwd, *a, **k):
            out = orig(fwd, *a, **k)
            rec.append(mac.actor_arena.grad.clone())
            return out
        ops.ac_backward = spy
    learner.probe_last_step = True
    learner.train(0)
    torch.cuda.synchronize()
    if light_spy:
        ops.ac_backward = orig
    return mac.actor_arena.grad.clone(), mac.actor_arena.data.clone(), rec, mac.actor_arena

g0, p0, _, ar = run(False)
g1, p1, _, _ = run(False)
g2, p2, rec, _ = run(True)
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
print("run0 vs run1 final grad:", rel(g0, g1), " params:", rel(p0, p1))
print("run0 vs light-spy run final grad:", rel(g0, g2))
print("light-spy run: final grad vs after-ac_backward grad (epoch 2):", rel(g2, rec[-1]))
for k in ("base.feature_norm.weight", "base.mlp.fc1.0.weight", "base.mlp.fc2.0.0.weight", "base.mlp.fc2.0.2.weight", "rnn.rnn.weight_ih_l0"):
    o, n = ar.offsets[k], int(torch.Size(ar.shapes[k]).numel())
    print(f"   {k}: run0 vs run1 {rel(g0[0, o:o+n], g1[0, o:o+n]):.2e}  final vs after-bwd {rel(g2[0, o:o+n], rec[-1][0, o:o+n]):.2e}  run0 vs spy-run {rel(g0[0, o:o+n], g2[0, o:o+n]):.2e}")
