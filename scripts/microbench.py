"""Per-piece timings of the hot path at BASELINE.json config 3 (E = 32, 5 agents x 55 entities), HIP events
around the host-level calls.  Usage: python scripts/microbench.py [piece ...]   (default: all)"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iplan_amd import ops  # noqa: E402
from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.harness import SyntheticLoop  # noqa: E402
from iplan_amd.nova.GAT_Net import gumbel_noise  # noqa: E402

E = int(os.environ.get("MB_ENVS", "32"))
args = default_args("highway", use_cuda=True, batch_size_run=E)
dev = torch.device("cuda")
loop = SyntheticLoop(args, E, seed=0, device=dev)
nA, N, d, Z, A = args.n_agents, args.max_vehicle_num, args.obs_shape_single, args.latent_dim, args.attention_dim
want = set(sys.argv[1:])


def tm(name, fn, n=10, warm=2):
    if want and name.split(":")[0] not in want:
        return
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:34s} gpu {e0.elapsed_time(e1) / n:9.3f} ms   wall {(time.perf_counter() - t0) / n * 1e3:9.3f} ms", flush=True)


hist = loop.obs_sets[0]["hist"][0].permute(1, 0, 2, 3)
lat = torch.softmax(torch.randn(nA, E, N, Z, device=dev), -1)
hid = torch.randn(nA, E, N, A, device=dev) * 0.1
noise = gumbel_noise((nA, E, N, N - 1, 2), dev)
out = torch.empty(nA, E, N, A, device=dev)
tm("gat_fwd", lambda: ops.gat_forward(loop.prediction.gat_arena, hist, lat, hid, noise, out=out), n=50)

window = loop.obs_sets[0]["hist"][0:10].permute(1, 2, 3, 0, 4).contiguous()
eh = torch.zeros(E, 1, nA, N, 32, device=dev)
lat_e = lat.permute(1, 0, 2, 3).contiguous()
tm("enc_fwd", lambda: loop.behavior.latent_update(window, eh, lat_e), n=50)

with contextlib.redirect_stdout(io.StringIO()):
    batch = loop.rollout()
tm("select_actions", lambda: loop.mac.select_actions_ippo(batch, 3, test_mode=False, as_numpy=False), n=50)
tm("rollout", lambda: loop.rollout(), n=2, warm=1)
if os.environ.get("MB_DEFER"):
    # the form the training cycle runs (harness.cycle): decoder weight gradients + optimiser step as ONE deferred iplan_wgrad call
    # on the side stream -- the PMC passes use it so that the per-kernel counters are those of the kernels bench.py times
    def beh_learn():
        loop.behavior.learn(batch, 0, defer_decoder=True)
        loop.behavior.join_decoder()
    tm("behavior_learn", beh_learn, n=3, warm=1)
else:
    tm("behavior_learn", lambda: loop.behavior.learn(batch, 0), n=3, warm=1)
tm("prediction_learn", lambda: loop.prediction.learn(batch, 0), n=5, warm=1)


def ppo():
    while not loop.learner.buffers[0].can_sample():
        loop.learner.insert_episode_batch(batch)
    with contextlib.redirect_stdout(io.StringIO()):
        loop.learner.train(0)


tm("ppo_train", ppo, n=2, warm=1)

if not want or "ac_train_parts" in want:
    # the pieces of one PPO epoch
    while not loop.learner.buffers[0].can_sample():
        loop.learner.insert_episode_batch(batch)
    L = loop.learner
    dd = L.store.data
    T = args.episode_limit
    acts = dd["actions"][..., 0]
    last = torch.cat([acts[:, :1], acts[:, :-1]], dim=1).to(torch.int32).contiguous()
    spec = L._feature_spec(T, T + 1, last)
    ha, hc = dd["rnn_states_actors"], dd["rnn_states_critics"]
    avail = dd["avail_actions"]
    rows = args.batch_size * T
    kw = dict(h_actor=ha, h_critic=hc, h_strides=(ha.stride(2), ha.stride(1)), avail=avail, avail_strides=(avail.stride(2), avail.stride(1)),
              mode=2, actions_in=dd["actions"], act_strides=(dd["actions"].stride(2), dd["actions"].stride(1)), n_actions=5, ksplit=1, want_h=False)
    want = set()
    tm("ac_fwd_train(infer)", lambda: ops.ac_forward(loop.mac.actor_arena, loop.mac.critic_arena, 2, spec, rows, nA, **kw), n=5)
    tm("ac_fwd_train(save)", lambda: ops.ac_forward(loop.mac.actor_arena, loop.mac.critic_arena, 2, spec, rows, nA, save=True, want_entropy=True, **kw), n=5)
    fo = ops.ac_forward(loop.mac.actor_arena, loop.mac.critic_arena, 2, spec, rows, nA, save=True, want_entropy=True, **kw)
    # what a PPO epoch launches: cached LayerNorm(F) statistics (mode 2) and the pre-packed fc1 operands
    lns = torch.empty(nA, args.batch_size * (T + 1), 2, device=dev)
    pk = loop.mac.fc1_pack.get(spec)
    ops.ac_forward(loop.mac.actor_arena, loop.mac.critic_arena, 2, spec, rows, nA, ln_stats=lns, ln_stats_mode=1, packed=pk, **kw)
    tm("ac_fwd_train(epoch form)", lambda: ops.ac_forward(loop.mac.actor_arena, loop.mac.critic_arena, 2, spec, rows, nA, save=True, want_entropy=True,
                                                          ln_stats=lns, ln_stats_mode=2, packed=pk, **kw), n=5)
    clk = torch.zeros(4, dtype=torch.int64, device=dev)
    for _ in range(2):
        ops.ac_forward(loop.mac.actor_arena, loop.mac.critic_arena, 2, spec, rows, nA, save=True, want_entropy=True, ln_stats=lns, ln_stats_mode=2,
                       packed=pk, phase_clocks=clk, **kw)
    torch.cuda.synchronize()
    cc = clk.cpu().double()
    print("ac_fwd (epoch form) phases of wave 0 of WG 0, x10 ns: stats, fc1 contraction, tail (2 row tiles) =", (cc[1:] - cc[:-1]).tolist())
    g1 = torch.randn(nA, rows, device=dev)
    tm("ac_backward(all)", lambda: ops.ac_backward(fo, loop.mac.actor_arena, loop.mac.critic_arena, g_logp=g1, g_entropy=-1e-6, g_values=g1), n=5)

if "gat_phases12" in set(sys.argv[1:]):           # libraries built with -DGAT_P3_CLOCKS only
    clk = torch.zeros(nA * E, 12, dtype=torch.int64, device=dev)
    ops.gat_forward(loop.prediction.gat_arena, hist, lat, hid, noise, out=out, phase_clocks=clk)
    torch.cuda.synchronize()
    c = clk.cpu().double()
    seq = c[:, [0, 1, 2, 5, 6, 7, 8, 9, 3, 4]]
    print("gat clocks x10 ns, mean per WG: entry->p1->p2->[noise issued]->[score GEMM]->[softmax]->[gate]->[aggregate]->barrier->p4:",
          (seq[:, 1:] - seq[:, :-1]).mean(0).tolist())

if "gat_bwd_phases" in set(sys.argv[1:]):     # training-form GAT (Prediction_policy.learn's shape: 64 sampled scenes per net)
    S = 64
    ar = loop.prediction.gat_arena
    ob = torch.rand(nA, S, N, hist.shape[-1], device=dev) * 2 - 1
    la = torch.rand(nA, S, N, lat.shape[-1], device=dev)
    hi = torch.randn(nA, S, N, 32, device=dev) * 0.1
    from iplan_amd.nova.GAT_Net import gumbel_noise
    nz = gumbel_noise((nA, S, N, N - 1, 2), dev)
    go = torch.randn(nA, S, N, 32, device=dev)

    def fb(clk=None):
        o, saved = ops.gat_forward(ar, ob, la, hi, nz, save=True)
        ops.gat_backward(ar, saved, go, phase_clocks=clk)
    tm("gat_fwd(save)+bwd+wgrad S=64", fb, n=5)
    tm("gat_fwd(save) S=64", lambda: ops.gat_forward(ar, ob, la, hi, nz, save=True), n=5)
    clk = torch.zeros(16, dtype=torch.int64, device=dev)
    fb(clk)
    torch.cuda.synchronize()
    c = clk.cpu().double()[8:15]
    print("gat_bwd phases of WG 0, x10 ns: A cell', B attention', C dk dv, D pair-GRU BPTT, E gather, F projections' =", (c[1:] - c[:-1]).tolist())

if "gat_phases" in set(sys.argv[1:]):
    clk = torch.zeros(nA * E, 5, dtype=torch.int64, device=dev)
    ops.gat_forward(loop.prediction.gat_arena, hist, lat, hid, noise, out=out, phase_clocks=clk)
    torch.cuda.synchronize()
    c = clk.cpu().double()
    d = (c[:, 1:] - c[:, :-1])
    print("gat phases (wall_clock64 ticks, 100 MHz => x10 ns): mean per WG", d.mean(0).tolist(), "max", d.max(0).values.tolist())
    print("kernel span ticks:", float(c[:, 4].max() - c[:, 0].min()), "first-start spread", float(c[:, 0].max() - c[:, 0].min()))

if "ac_phases" in set(sys.argv[1:]):
    clk = torch.zeros(4, dtype=torch.int64, device=dev)
    for _ in range(3):
        loop.mac.select_actions_ippo(batch, 3, test_mode=False, as_numpy=False, phase_clocks=clk)
    torch.cuda.synchronize()
    c = clk.cpu().double()
    print("ac_fwd (rollout) phases of WG 0, x10 ns: stats, contraction, tail =", (c[1:] - c[:-1]).tolist())

if "defer_overlap" in set(sys.argv[1:]):
    # critical path of Behavior_policy.learn with the decoder update in line / deferred, and the rollout that follows it
    def seq(defer, n=4):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tl = tr = 0.0
        for it in range(n + 1):
            torch.cuda.synchronize()
            ev[0].record()
            loop.behavior.learn(batch, 0, **({"defer_decoder": True} if defer else {}))
            ev[1].record()
            loop.rollout()
            ev[2].record()
            torch.cuda.synchronize()
            if it:
                tl += ev[0].elapsed_time(ev[1]) / n
                tr += ev[1].elapsed_time(ev[2]) / n
        return tl, tr
    work = torch.cuda.Stream(dev)          # (CU-masked streams are blocking streams: keep off the legacy default stream)
    for defer in (False, True, True):
        with torch.cuda.stream(work):
            tl, tr = seq(defer)
        print(f"defer_decoder={defer}: learn (main stream) {tl:.2f} ms, following rollout {tr:.2f} ms, sum {tl + tr:.2f} ms")

if "masked_wgrad" in set(sys.argv[1:]):
    # the deferred decoder update alone (nothing else on the GPU) on streams restricted to k CUs
    from iplan_amd.streams import masked_stream
    for k in (0, 128, 96, 64):
        loop.behavior._dec_stream = masked_stream(dev, k) if k else torch.cuda.Stream(dev)
        ts = []
        for it in range(3):
            loop.behavior.join_decoder()
            torch.cuda.synchronize()
            loop.behavior.learn(batch, 0, defer_decoder=True)
            torch.cuda.current_stream().synchronize()
            t0 = time.perf_counter()
            loop.behavior._dec_stream.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"deferred decoder update on {k or 256} CUs: {min(ts[1:]):.2f} ms after learn() returned (host clock)")
