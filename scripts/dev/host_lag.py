"""Is the head of the learn phase host-bound?  (no profiler: rocprofv3 inflates every launch's host cost)

A training cycle at config 3, 8 rollouts per step.  Events on the rollout's stream: R0 at the rollout's first enqueue, R1 behind its
last launch (+ insert + prepare_learn).  At chosen host points of the cycle the script notes the host clock and whether R1 has
already completed on the device (= the device has nothing of this phase queued yet: the host is behind).  It also times, with
events, from R1 to the behaviour decoder forward's first launch.

    python scripts/dev/host_lag.py            (env knobs of the harness apply: IPLAN_BEH_FIRST, IPLAN_RUN_AHEAD, IPLAN_ROLLOUT_GRAPH)
"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from iplan_amd import ops  # noqa: E402
from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.harness import SyntheticLoop  # noqa: E402

E = 32
args = default_args("highway", use_cuda=True, batch_size_run=E)
dev = torch.device("cuda")
loop = SyntheticLoop(args, E, seed=1234, device=dev)
log = []          # per cycle: dict of host stamps / queries
cur = {}


def stamp(name, ev=None):
    cur[name] = (time.perf_counter(), None if ev is None else ev.query())


orig_rollout = loop.rollout
orig_prep = loop.behavior.prepare_learn
orig_pl = loop.prediction.learn
orig_bl = loop.behavior.learn
orig_train = loop.learner.train


def rollout():
    cur.clear()
    cur["R0"] = torch.cuda.Event(enable_timing=True)
    cur["R0"].record()
    stamp("rollout_enq_start")
    b = orig_rollout()
    cur["R1r"] = torch.cuda.Event(enable_timing=True)
    cur["R1r"].record()
    stamp("rollout_enq_end", cur["R1r"])
    return b


def prep(batch):
    r = orig_prep(batch)
    cur["R1"] = torch.cuda.Event(enable_timing=True)
    cur["R1"].record()
    stamp("prepared", cur["R1"])
    return r


def pl(*a, **k):
    stamp("pred_learn_start", cur["R1"])
    r = orig_pl(*a, **k)
    cur["P1"] = torch.cuda.Event(enable_timing=True)
    cur["P1"].record()                       # (current stream = the side stream the cycle runs prediction learning on)
    stamp("pred_learn_enqueued", cur["R1"])
    return r


def train(*a, **k):
    stamp("train_start", cur["R1"])
    r = orig_train(*a, **k)
    cur["T1"] = torch.cuda.Event(enable_timing=True)
    cur["T1"].record()
    stamp("train_enqueued", cur["R1"])
    return r


def bl(*a, **k):
    stamp("beh_learn_start", cur["R1"])
    cur["B0"] = torch.cuda.Event(enable_timing=True)
    cur["B0"].record()                       # on the main stream: completes when the device reaches the head of behaviour learn
    r = orig_bl(*a, **k)
    cur["B1"] = torch.cuda.Event(enable_timing=True)
    cur["B1"].record()
    stamp("beh_learn_returned", cur["R1"])
    return r


orig_cycle = loop.cycle


def cycle():
    n = orig_cycle()
    if "B1" in cur:
        stamp("cycle_returned")
        log.append(dict(cur))
    return n


loop.rollout = rollout
loop.behavior.prepare_learn = prep
loop.prediction.learn = pl
loop.behavior.learn = bl
loop.learner.train = train
if not loop._defer_cus:
    loop.cycle = cycle

with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(2 * 8):
        loop.cycle()
    torch.cuda.synchronize()
    log.clear()
    t0 = time.perf_counter()
    for _ in range(3 * 8):
        loop.cycle()
    loop.finish()
    torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per 8-rollout step")
names = ["rollout_enq_start", "rollout_enq_end", "prepared", "pred_learn_start", "pred_learn_enqueued", "train_start", "train_enqueued",
         "beh_learn_start", "beh_learn_returned", "cycle_returned"]
names = [n for n in names if all(n in c for c in log)]
names.sort(key=lambda n: sum(c[n][0] - c["rollout_enq_start"][0] for c in log))
print("host point              ms after the rollout's first enqueue (mean)    device already past the rollout's end (share of cycles)")
for n in names:
    dt = sum(c[n][0] - c["rollout_enq_start"][0] for c in log) / len(log) * 1e3
    q = [c[n][1] for c in log if c[n][1] is not None]
    print(f"{n:24s}{dt:9.3f}" + (f"{sum(q) / len(q):42.2f}" if q else ""))
print("device: rollout R0->R1r %.3f ms, R1r->R1 (insert + prepare) %.3f ms, R1->B0 (main stream reaches behaviour learn) %.3f ms, "
      "B0->B1 (behaviour learn on the main stream) %.3f ms" % tuple(
          sum(c[a].elapsed_time(c[b]) for c in log) / len(log) for a, b in (("R0", "R1r"), ("R1r", "R1"), ("R1", "B0"), ("B0", "B1"))))
ft = [c for c in log if c["R1"].elapsed_time(c["T1"]) >= 5.0]        # cycles whose train() acted (buffer full)
print("device: R1->P1 (prediction learn done) %.3f ms; R1->T1 (PPO train done; buffer-full cycles) %.3f ms; B1->T1 there %.3f ms" % (
    sum(c["R1"].elapsed_time(c["P1"]) for c in log) / len(log), sum(c["R1"].elapsed_time(c["T1"]) for c in ft) / max(1, len(ft)),
    sum(c["B1"].elapsed_time(c["T1"]) for c in ft) / max(1, len(ft))))
fid = {id(c) for c in ft}
plain = [a["B1"].elapsed_time(b["R0"]) for a, b in zip(log, log[1:]) if id(a) not in fid]
after = [a["B1"].elapsed_time(b["R0"]) for a, b in zip(log, log[1:]) if id(a) in fid]
print("device: B1 -> the next cycle's R0: %.3f ms (no PPO update in between), %.3f ms (behind a PPO update)" % (
    sum(plain) / max(1, len(plain)), sum(after) / max(1, len(after))))
