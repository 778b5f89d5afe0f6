"""CPU, world_size 2, gloo: the data-parallel exchange step.  Each rank runs the learners on its own
shard (host-emulated kernels); after the gradient all-reduce the replicas must hold bit-identical
parameters, and the averaged gradient must equal the mean of the two single-rank gradients."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, io, contextlib
sys.path.insert(0, os.environ["IPLAN_ROOT"])
import torch
import torch.distributed as dist
from iplan_amd import _lib as L
from tests.emu.emu_lib import get_emu_lib
L.use_library_for_tests(get_emu_lib())
from iplan_amd.config import default_args
from iplan_amd.harness import SyntheticLoop
from iplan_amd.parallel import DataParallel
dist.init_process_group("gloo")
rank = dist.get_rank()
args = default_args("highway", use_cuda=False, max_vehicle_num=4, n_agents=2, episode_limit=12, batch_size_run=2,
                    buffer_size=2, batch_size=1, ppo_epoch=2, pred_batch_size=4, max_history_len=3)
loop = SyntheticLoop(args, 2, seed=100 + rank, device="cpu")       # different data AND different initial weights per rank
dp = DataParallel().attach(loop)
arenas = [loop.mac.actor_arena, loop.mac.critic_arena, loop.behavior.enc_arena, loop.behavior.dec_arena,
          loop.prediction.gat_arena, loop.prediction.dec_arena]
def gathered(t):
    out = [torch.empty_like(t) for _ in range(2)]
    dist.all_gather(out, t.contiguous())
    return out
for a in arenas:                                                    # broadcast made the replicas identical
    g = gathered(a.data)
    assert torch.equal(g[0], g[1])
# averaged gradient == mean of the per-rank gradients (check on the behaviour nets)
calls = []
orig = dp.all_reduce_grads
def spy(*ar):
    before = [gathered(a.grad) for a in ar]
    orig(*ar)
    for a, b in zip(ar, before):
        assert torch.allclose(a.grad, (b[0] + b[1]) / 2, rtol=0, atol=1e-7)
    calls.append(len(ar))
dp.all_reduce_grads = spy
with contextlib.redirect_stdout(io.StringIO()):
    loop.cycle()
assert len(calls) == 1 + 1 + args.ppo_epoch, calls               # behaviour, prediction, one per PPO epoch
for a in arenas:                                                    # replicas still bit-identical after all updates
    g = gathered(a.data)
    assert torch.equal(g[0], g[1]), "replicas diverged"
    assert torch.isfinite(a.data).all()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_data_parallel_two_ranks_gloo(tmp_path):
    from tests.emu.emu_lib import get_emu_lib
    get_emu_lib()                                               # build the emulated library once, before the ranks race for it
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, IPLAN_ROOT=ROOT, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2
