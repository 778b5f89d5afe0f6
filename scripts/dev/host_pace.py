"""Diagnostic: is the rollout's launch sequence host-bound?  Host time to ENQUEUE a rollout (no synchronisation) against the
GPU time it takes, and the same for the learn phase's enqueue."""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.harness import SyntheticLoop  # noqa: E402

args = default_args("highway", use_cuda=True, batch_size_run=32)
loop = SyntheticLoop(args, 32, seed=0, device=torch.device("cuda"))
for _ in range(2):
    loop.rollout()
torch.cuda.synchronize()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    b = loop.rollout()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rollout: host enqueue {1e3 * (t1 - t0):7.2f} ms, until the GPU is done {1e3 * (t2 - t0):7.2f} ms, GPU span {e0.elapsed_time(e1):7.2f} ms", flush=True)
with contextlib.redirect_stdout(io.StringIO()):
    loop.cycle()
torch.cuda.synchronize()
for rep in range(3):
    b = loop.rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.learner.insert_episode_batch(b)
    t1 = time.perf_counter()
    f = loop.prediction.learn(b, 0, defer=True)
    t2 = time.perf_counter()
    g = loop.behavior.learn(b, 0, defer_decoder=True, defer_readback=True)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    f(); g()
    print(f"learn phase enqueue: insert {1e3 * (t1 - t0):6.2f} ms, prediction.learn {1e3 * (t2 - t1):6.2f} ms, behaviour.learn {1e3 * (t3 - t2):6.2f} ms; GPU done after {1e3 * (t4 - t0):6.2f} ms", flush=True)
