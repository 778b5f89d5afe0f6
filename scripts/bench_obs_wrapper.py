"""CPU-baseline leg for SURVEY.md §8f.2 (like bench.py's cpu_baseline it may time the oracle; nothing here is product code).
Host cost of the id -> slot history wrapper per vector step at BASELINE config 3 (32 threads, 5 agents, 15 observed
rows, 55 slots, L = 10): the loop restatement of the reference (oracle, = the reference's own cost) vs the vectorised
iplan_amd.observation_wrapper.  Pure CPU; python scripts/bench_obs_wrapper.py"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iplan_amd.observation_wrapper import observersation_state_history_wrapper as Wrapper  # noqa: E402
from iplan_amd.synth import obs_stream  # noqa: E402
from oracle.obs_wrapper_oracle import HistoryWrapperOracle  # noqa: E402

K, nA, obs_num, d, T, L, N = 32, 5, 15, 5, 90, 10, 55
steps = obs_stream(K, nA, obs_num, d, T, seed=0, n_ids=40)
w = Wrapper(SimpleNamespace(obs_shape_single=d, batch_size_run=K), nA, N, T, L)
o = HistoryWrapperOracle(K, nA, N, T, L, d)
w.agent_obs_profile_init(steps[0])
o.init(steps[0])
t0 = time.perf_counter()
for obs in steps:
    w.obs_history_create(obs)
    a = w.obs_single_history_output()
    b = w.obs_history_output()
t_vec = (time.perf_counter() - t0) / T
t0 = time.perf_counter()
for obs in steps:
    o.create(obs)
    a2 = o.single()
    b2 = o.window(L)
t_loop = (time.perf_counter() - t0) / T
assert np.array_equal(a, a2) and np.array_equal(b, b2)
print(f"per vector step (create + single + window outputs), {K} threads x {nA} agents x {obs_num} rows: "
      f"loop restatement {t_loop * 1e3:.1f} ms, vectorised {t_vec * 1e3:.2f} ms  ({t_loop / t_vec:.0f}x)")
