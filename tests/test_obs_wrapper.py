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


# ------------------------------------------------------------------------------------------------ the device-resident form
def check_device_history(g, device):
    """iplan_obs_history_step (csrc/obs_history.hip) against the reference-recorded fixture, step by step: windows and single-step
    views bit-identical to the reference's after the float32 cast the runner applies anyway; slot bookkeeping identical; the
    single-step view written through strides into a larger container"""
    from iplan_amd.observation_wrapper import DeviceObsHistory
    D = g["dims"]
    dh = DeviceObsHistory(D["K"], D["nA"], D["N"], D["L"], D["d"], device)
    dh.init(g["steps"][0])
    T = len(g["steps"])
    # the target of the single-step view: a time slice of an episode container (strided over threads and agents)
    dense = torch.zeros(D["K"], T + 1, D["nA"], D["N"], D["d"], device=device)
    for t, obs in enumerate(g["steps"]):
        win = dh.step(obs, single_out=dense[:, t])
        dh.stage_error_flag()
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize()
        dh.check()
        assert torch.equal(win.cpu(), torch.as_tensor(g["hist"][t]).float()), t
        assert torch.equal(dense[:, t].cpu(), torch.as_tensor(g["single"][t]).float()), t
    ids = dh.slot_id.cpu().numpy()
    ns = dh.n_slots.cpu().numpy()
    got = [[[int(x) for x in ids[k, i, :ns[k, i]]] for i in range(D["nA"])] for k in range(D["K"])]
    assert got == g["vehicle_ids"]


def check_device_history_vs_numpy_stream(device, K=6, nA=3, obs_num=7, d=5, N=12, L=4, T=25, dup=False):
    """a longer stub-simulator stream (vehicles come and go) against the numpy class, incl. two agents of a thread that report the
    SAME ego id (both feed one state, each with its own zero entries).  The duplicate case is device-vs-NUMPY only: the two walk
    rows-outer / agents-inner, the reference walks agent-outer (observation_wrapper.py:92-96) and itself raises IndexError in
    obs_history_output with a duplicate ego id, so there is no reference answer to compare with."""
    from iplan_amd import synth
    from iplan_amd.observation_wrapper import DeviceObsHistory, observersation_state_history_wrapper as Wrapper
    env = synth.StubHighwayVecEnv(K, nA, obs_num, d, N, T, seed=3, n_ids=N - nA - 1)
    w = Wrapper(SimpleNamespace(obs_shape_single=d, batch_size_run=K), nA, N, T + 2, L)
    dh = DeviceObsHistory(K, nA, N, L, d, device)
    _, obs = env.reset()
    if dup:
        env.obs[:, 1, 2, 0, 0] = env.obs[:, 1, 0, 0, 0]            # thread 1: agent 2 reports agent 0's ego id
        obs = env.obs[0]
    w.agent_obs_profile_init(obs)
    dh.init(obs)
    single = torch.zeros(K, nA, N, d, device=device)
    for t in range(T):
        w.obs_history_create(obs)
        win = dh.step(obs, single_out=single)
        assert torch.equal(win.cpu(), torch.as_tensor(w.obs_history_output()).float()), t
        assert torch.equal(single.cpu(), torch.as_tensor(w.obs_single_history_output()).float()), t
        _, obs, *_ = env.step([None] * K)
    dh.stage_error_flag()
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    dh.check()


@pytest.fixture
def emu_lib():
    from iplan_amd import _lib as L
    from tests.emu.emu_lib import get_emu_lib
    L.use_library_for_tests(get_emu_lib())
    yield
    L.use_library_for_tests(None)


def test_device_history_matches_reference_emulated(g, emu_lib):
    check_device_history(g, "cpu")


@pytest.mark.parametrize("dup", [False, True])
def test_device_history_vs_numpy_stream_emulated(emu_lib, dup):
    check_device_history_vs_numpy_stream("cpu", dup=dup)


def test_device_history_errors_emulated(g, emu_lib):
    from iplan_amd.observation_wrapper import DeviceObsHistory
    D = g["dims"]
    dh = DeviceObsHistory(D["K"], D["nA"], D["N"], D["L"], D["d"], "cpu")
    dh.init(g["steps"][0])
    bad = np.array(g["steps"][1], copy=True)
    bad[1, 0, 0, 0] = 9999
    dh.step(bad)
    with pytest.raises(ValueError):
        dh.check()
    small = DeviceObsHistory(D["K"], D["nA"], 2, D["L"], D["d"], "cpu")
    small.init(g["steps"][0])
    with pytest.raises(IndexError):
        for obs in g["steps"]:
            small.step(obs)
            small.check()


@pytest.mark.gpu
def test_device_history_matches_reference_gpu(g):
    check_device_history(g, "cuda")
    check_device_history_vs_numpy_stream("cuda", K=32, nA=5, obs_num=15, d=5, N=55, L=10, T=40)
    check_device_history_vs_numpy_stream("cuda", dup=True)
