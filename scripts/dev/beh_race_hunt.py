"""Which kernel of Behavior_policy.learn is not bit-reproducible?  (1 learn in ~600 differs from the first by ~1e-6 of a gradient's
max: profiles/r06_notes.md section 9.)  The forward (encoder + decoder ranges), the BPTT and the weight-gradient contraction are repeated
separately on the SAME inputs and every output compared bit for bit with the first run's.
    python scripts/dev/beh_race_hunt.py [reps] [stage: fwd|bwd|both]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from iplan_amd import ops, synth  # noqa: E402
from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.nova.stable_behavior_policy import Behavior_policy  # noqa: E402


class Log:
    def log_stat(self, *a, **k):
        pass


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
stage = sys.argv[2] if len(sys.argv) > 2 else "both"
args = default_args("highway", use_cuda=True, batch_size_run=32)
E = 32
f = synth.make_episode_fields(args, E, 34, 0.8)
batch = synth.DictBatch(f, E, args.episode_limit + 1).to("cuda")
nA, N, Lw, T = args.n_agents, args.max_vehicle_num, args.max_history_len, args.episode_limit
J = T - 1 - Lw
gen = torch.Generator().manual_seed(35)
keep = (torch.rand(nA, J, E * N, Lw, args.decoder_rnn_dim, generator=gen) < 0.9).to(torch.uint8).cuda()
torch.manual_seed(33)
pol = Behavior_policy(args, Log())
prep = pol.prepare_learn(batch)


def forward():
    return ops.beh_forward(pol.enc_arena, pol.dec_arena, prep["hist"], prep["mask"], pol.max_history_len, pol.latent_dim, pol.soft_update_coef,
                           pol.thres_small_variation, args.decoder_dropout, keep=keep, seed=0, win_norm=prep["win_norm"])


def describe(name, a, b):
    fa, fb = a.reshape(-1), b.reshape(-1)
    n, first = 0, []
    CH = 1 << 28
    for lo in range(0, fa.numel(), CH):
        x, y = fa[lo:lo + CH], fb[lo:lo + CH]
        d = (x != y) & ~(torch.isnan(x) & torch.isnan(y))
        c = int(d.sum())
        n += c
        if c and len(first) < 4:
            first += (torch.nonzero(d)[:4 - len(first), 0] + lo).tolist()
    shp = tuple(a.shape)
    where = [tuple(int(v) for v in torch.unravel_index(torch.tensor(i), shp)) for i in first]
    vals = [(float(fa[i]), float(fb[i])) for i in first]
    return f"{name}: {n} of {a.numel()} elements differ, shape {shp}, first at {where} values (first run, this run) {vals}"


FWD_KEYS = ("loss", "saved_dec", "saved_enc", "saved_lat")
ref = forward()
torch.cuda.synchronize()
ref_c = {k: ref[k].clone() for k in FWD_KEYS}
bad_f = 0
if stage in ("fwd", "both"):
    for r in range(reps):
        out = forward()
        torch.cuda.synchronize()
        for k in FWD_KEYS:
            if not torch.equal(out[k], ref_c[k]):
                bad_f += 1
                print("FWD rep", r, describe(k, ref_c[k], out[k]), flush=True)
                if k == "saved_dec":
                    # the differing (net, tile, group, step): what did this run store there, next to the first run's neighbouring steps?
                    fa, fb = ref_c[k].reshape(-1), out[k].reshape(-1)
                    CH = 1 << 28
                    for lo in range(0, fa.numel(), CH):
                        d = fa[lo:lo + CH] != fb[lo:lo + CH]
                        if bool(d.any()):
                            i0 = int(torch.nonzero(d)[0, 0]) + lo
                            break
                    net, tile, cg, step, ch, col = (int(v) for v in torch.unravel_index(torch.tensor(i0), ref_c[k].shape))
                    print(f"   at net {net} tile {tile} group {cg} step {step} (window {step // Lw}, t {step % Lw}); chain 0, columns 0..4:")
                    for st in range(max(step - 4, 0), min(step + 5, ref_c[k].shape[3])):
                        print(f"     step {st:4d}  first run {[round(float(v), 5) for v in ref_c[k][net, tile, cg, st, 0, :5]]}  this run "
                              f"{[round(float(v), 5) for v in out[k][net, tile, cg, st, 0, :5]]}", flush=True)
                    bad_steps = torch.nonzero((ref_c[k][net, tile, cg] != out[k][net, tile, cg]).any(-1).any(-1))[:, 0].tolist()
                    print("   differing steps of that (net, tile, group):", bad_steps, " windows pieces:", os.environ.get("IPLAN_BEH_PIECES_FWD", "default"))
        del out
    print(f"forward: {bad_f} tensor mismatches in {reps} repetitions", flush=True)

bad_b = 0
if stage in ("bwd", "both"):
    def backward():
        pol.enc_arena.grad.zero_()
        pol.dec_arena.grad.zero_()
        b = ops.beh_backward(pol.enc_arena, pol.dec_arena, ref, penalty=pol.behavior_variation_penalty, E_norm=E)
        torch.cuda.synchronize()
        return dict(dsave_dec=b["dsave_dec"], dsave_lat=b["dsave_lat"], enc_grad=pol.enc_arena.grad.clone(), dec_grad=pol.dec_arena.grad.clone())
    b0 = backward()
    b0 = {k: v.clone() for k, v in b0.items()}
    for r in range(reps):
        b = backward()
        for k in b0:
            if not torch.equal(b[k], b0[k]):
                bad_b += 1
                print("BWD rep", r, describe(k, b0[k], b[k]), flush=True)
        del b
    print(f"backward (+ in-line weight gradients): {bad_b} tensor mismatches in {reps} repetitions", flush=True)
