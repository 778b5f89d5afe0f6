"""id -> slot history wrapper: the vectorised iplan_amd.observation_wrapper and the loop oracle both
reproduce the reference class, via the fixture oracle/make_golden.py recorded from it (tests/golden/obs_wrapper.pt):
outputs after every step of a stream with vehicles coming and going, the masked episode output, the id bookkeeping."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "obs_wrapper.pt")


@pytest.fixture(scope="module")
def g():
    return torch.load(GOLD, weights_only=False)


def test_vectorised_wrapper_matches_reference(g):
    from iplan_amd.observation_wrapper import observersation_state_history_wrapper as Wrapper
    D = g["dims"]
    w = Wrapper(SimpleNamespace(obs_shape_single=D["d"], batch_size_run=D["K"]), D["nA"], D["N"], D["T"], D["L"])
    w.agent_obs_profile_init(g["steps"][0])
    for t, obs in enumerate(g["steps"]):
        agent_id, vehicle_id, history = w.obs_history_create(obs)
        assert np.array_equal(w.obs_history_output(), g["hist"][t]), t
        assert np.array_equal(w.obs_single_history_output(), g["single"][t]), t
    assert [[list(map(int, x)) for x in y] for y in vehicle_id] == g["vehicle_ids"]
    assert [list(map(int, x)) for x in agent_id] == g["agent_ids"]
    raw, seg = w.obs_history_episode_output(g["mask"])
    assert np.array_equal(raw, g["raw"]) and np.array_equal(seg, g["seg"])
    # the dict-of-deques view the reference returns
    k, i = 1, 0
    assert len(history[k][i]) == len(g["vehicle_ids"][k][i])
    ns, no = w.pure_obs_state_wrapper(g["state"], g["steps"][-1])
    assert np.array_equal(ns, g["new_state"]) and np.array_equal(no, g["new_obs"])


def test_deque_overflow_and_capacity(g):
    """More steps than max_episode_len: a full deque drops its oldest entry (deque(maxlen)); more ids than slots: IndexError
    like the reference's output arrays."""
    from iplan_amd.observation_wrapper import observersation_state_history_wrapper as Wrapper
    from oracle.obs_wrapper_oracle import HistoryWrapperOracle
    D = g["dims"]
    Tm = 5
    w = Wrapper(SimpleNamespace(obs_shape_single=D["d"], batch_size_run=D["K"]), D["nA"], D["N"], Tm, D["L"])
    o = HistoryWrapperOracle(D["K"], D["nA"], D["N"], Tm, D["L"], D["d"])
    w.agent_obs_profile_init(g["steps"][0])
    o.init(g["steps"][0])
    for obs in g["steps"]:
        w.obs_history_create(obs)
        o.create(obs)
        assert np.array_equal(w.obs_history_output(), o.window(D["L"]))
    small = Wrapper(SimpleNamespace(obs_shape_single=D["d"], batch_size_run=D["K"]), D["nA"], 2, D["T"], D["L"])
    small.agent_obs_profile_init(g["steps"][0])
    with pytest.raises(IndexError):
        for obs in g["steps"]:
            small.obs_history_create(obs)


def test_loop_oracle_matches_reference(g):
    from oracle.obs_wrapper_oracle import HistoryWrapperOracle
    D = g["dims"]
    o = HistoryWrapperOracle(D["K"], D["nA"], D["N"], D["T"], D["L"], D["d"])
    o.init(g["steps"][0])
    for t, obs in enumerate(g["steps"]):
        o.create(obs)
        assert np.array_equal(o.window(D["L"]), g["hist"][t]) and np.array_equal(o.single(), g["single"][t])
    assert np.array_equal(o.window(D["T"], g["mask"]), g["raw"])


def test_unregistered_ego_id_raises_like_the_reference(g):
    """observation_wrapper.py:76 does ``self.agent_id[k].index(agent_id)``: an ego id agent_obs_profile_init never saw is a
    ValueError, not a silent write into agent slot 0."""
    from iplan_amd.observation_wrapper import observersation_state_history_wrapper as Wrapper
    D = g["dims"]
    w = Wrapper(SimpleNamespace(obs_shape_single=D["d"], batch_size_run=D["K"]), D["nA"], D["N"], D["T"], D["L"])
    w.agent_obs_profile_init(g["steps"][0])
    bad = np.array(g["steps"][1], copy=True)
    bad[1, 0, 0, 0] = 9999
    with pytest.raises(ValueError):
        w.obs_history_create(bad)
