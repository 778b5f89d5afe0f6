"""Exercise the data-parallel code path (RCCL process group, gradient all-reduces on the learner streams) with a
1-rank group on one GPU: 9 full cycles (so that IPPOLearner.train runs once), then parameters must be finite.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 scripts/dp_single_rank_check.py"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.harness import SyntheticLoop  # noqa: E402
from iplan_amd.parallel import DataParallel  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
E = int(os.environ.get("DP_ENVS", "32"))
args = default_args("highway", use_cuda=True, batch_size_run=E)
loop = SyntheticLoop(args, E, seed=7, device=dev)
DataParallel(dist.group.WORLD).attach(loop)
n = max(1, args.buffer_size // E) + 1
torch.cuda.synchronize()
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(n):
        loop.cycle()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for a in (loop.mac.actor_arena, loop.mac.critic_arena, loop.behavior.enc_arena, loop.behavior.dec_arena,
          loop.prediction.gat_arena, loop.prediction.dec_arena):
    assert torch.isfinite(a.data).all()
print(f"dp single-rank check ok: {n} cycles in {dt:.2f} s ({n * E * args.episode_limit / dt:.0f} env-steps/s incl. first-call overheads)")
dist.destroy_process_group()
