"""CPU, world_size 2, gloo: the data-parallel exchange step (host-emulated kernels).

Two things are checked on both ranks:
* EXACTNESS -- a 2-rank step equals the step of ONE process that holds the union of the ranks' data: every learner
  (behaviour, prediction, PPO) is run once on the full 4-env batch in a single-process replica and once data-parallel on
  the rank's 2-env half; the post-step parameters must agree (global loss normalisers + summed gradients);
* the replicas stay bit-identical through a full synthetic training cycle."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = os.path.join(ROOT, "tests", "dp_worker.py")        # shared with tests/test_dp_gpu_cycle.py


def test_data_parallel_two_ranks_gloo(tmp_path):
    from tests.emu.emu_lib import get_emu_lib
    get_emu_lib()                                               # build the emulated library once, before the ranks race for it
    env = dict(os.environ, IPLAN_ROOT=ROOT, OMP_NUM_THREADS="2", IPLAN_DP_DEVICE="cpu")
    env.pop("IPLAN_P2P_ALLREDUCE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", WORKER],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2
