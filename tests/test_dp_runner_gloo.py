"""CPU, world_size 2, gloo: the run_ippo-side data-parallel recipe -- ``parallel.init_from_env`` +
``parallel.shard_args`` + ``DataParallel.attach(mac=, learner=, behavior=, prediction=, runner=)`` around the objects
``run_ippo.run_sequential`` builds and the device-resident ``ParallelRunner`` on each rank's env shard -- equals ONE process
on the union (tests/dp_runner_worker.py says what is compared).  Reference loop: run_ippo.py:261-332,
runners/ippo_parallel_runner.py:105-281."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_runner_worker.py")


def test_run_ippo_recipe_two_ranks_gloo():
    from tests.emu.emu_lib import get_emu_lib
    get_emu_lib()                                               # build the emulated library once, before the ranks race for it
    env = dict(os.environ, IPLAN_ROOT=ROOT, OMP_NUM_THREADS="2")
    env.pop("IPLAN_P2P_ALLREDUCE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29633", WORKER],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


def test_shard_args_strong_and_weak():
    from iplan_amd.config import default_args
    from iplan_amd.parallel import shard_args
    import pytest
    u = default_args("highway", batch_size_run=256, buffer_size=256, batch_size=255)
    for r in range(8):
        a = shard_args(u, 8, r)
        assert (a.batch_size_run, a.buffer_size) == (32, 32)
        assert a.batch_size == (31 if r == 7 else 32)
        assert (a.dp_global_rows, a.dp_global_count) == (255 * 90, 256 * 90)
    assert sum(shard_args(u, 8, r).batch_size for r in range(8)) == u.batch_size
    assert u.batch_size_run == 256 and not hasattr(u, "dp_world")          # the caller's namespace is left alone
    w = shard_args(u, 8, 3, "weak")
    assert (w.batch_size_run, w.buffer_size, w.batch_size, w.dp_global_rows) == (256, 256, 255, None)
    with pytest.raises(ValueError):
        shard_args(default_args("highway", batch_size_run=30), 8, 0)
