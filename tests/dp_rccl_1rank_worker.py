"""TEST INFRASTRUCTURE (run by tests/test_dp_gpu_cycle.py in a process of its own): the DEFAULT collective path -- RCCL through
torch.distributed, ``dist.all_reduce(async_op=True)`` issued from the training cycle's four producer streams
(parallel.DataParallel.all_reduce_grads / all_reduce_sum) -- on the only RCCL group a single-GPU box can form: ONE rank.  A
1-rank sum all-reduce is the identity, so three full multi-stream cycles with the data-parallel hooks attached must leave all
six parameter arenas BIT-IDENTICAL to the same three cycles without them: what is checked is the stream ordering of
parallel.py's collectives against real HIP streams (a collective that ran before its producer finished, or a consumer that
ran before the collective, changes bits), not the arithmetic.  Prints "rccl-1rank ok" on success."""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.environ.get("IPLAN_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.harness import SyntheticLoop  # noqa: E402
from iplan_amd.parallel import DataParallel  # noqa: E402


def arenas_of(lp):
    return [lp.mac.actor_arena, lp.mac.critic_arena, lp.behavior.enc_arena, lp.behavior.dec_arena, lp.prediction.gat_arena, lp.prediction.dec_arena]


def run(args, E, attach):
    lp = SyntheticLoop(args, E, seed=300, device="cuda")
    if attach:
        DataParallel(dist.group.WORLD).attach(lp)
        assert lp.learner.dp is not None and lp.behavior.dp is not None and lp.prediction.dp is not None
        assert lp.learner.dp.backend == "nccl" and not lp.learner.dp.use_p2p
    for c in range(3):
        torch.manual_seed(1000 + c)
        torch.cuda.manual_seed(1000 + c)
        np.random.seed(1000 + c)
        with contextlib.redirect_stdout(io.StringIO()):
            lp.cycle()                                       # buffer_size == E: IPPOLearner.train acts in every cycle
    lp.finish()
    lp.behavior.join_decoder()
    torch.cuda.synchronize()
    return [a.data.clone() for a in arenas_of(lp)]


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29900 + os.getpid() % 90}", rank=0, world_size=1, device_id=dev)
    names = ["actor", "critic", "beh_enc", "beh_dec", "gat", "pred_dec"]
    for tag, kw, E in (("mid", dict(max_vehicle_num=9, n_agents=2, episode_limit=20, ppo_epoch=3, pred_batch_size=8), 4),
                       ("cfg3-width", dict(episode_limit=14, ppo_epoch=2, pred_batch_size=16), 8)):
        args = default_args("highway", use_cuda=True, batch_size_run=E, buffer_size=E, batch_size=E, **kw)
        plain = run(args, E, attach=False)
        hooked = run(args, E, attach=True)
        diffs = {n: float((a - b).abs().max()) for n, a, b in zip(names, plain, hooked)}
        assert all(torch.equal(a, b) for a, b in zip(plain, hooked)), (tag, diffs)
        moved = [float((a - b).abs().max()) for a, b in zip(plain, run(args, E, attach=False))]
        assert all(m == 0.0 for m in moved), ("the un-hooked cycle itself is not reproducible", tag, moved)
        print(tag, "hooked == plain, bit for bit", flush=True)
    dist.destroy_process_group()
    print("rccl-1rank ok")


if __name__ == "__main__":
    main()
