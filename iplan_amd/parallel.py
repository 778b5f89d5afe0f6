"""Data-parallel training across the GPUs of a node: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), every rank owns its shard of the parallel environments and a full replica of
all networks.

The reference has no multi-GPU path at all; this is the MI355X-native addition the
north star asks for.  Rollout inference and the three learners' forward/backward run rank-local with
no communication; the only exchange step is ONE sum all-reduce per gradient arena per optimiser step
(PPO: actor + critic arenas, 15x per train(); prediction: GAT + decoder arenas once per rollout;
behaviour: encoder + decoder arenas once per rollout) -- each arena is a single contiguous fp32
buffer holding all agents' gradients, so there are no per-tensor collectives.  Messages are 0.3-4 MB:
latency-bound on the xGMI mesh, hence as few and as large as the algorithm allows.  Gradients are
averaged over ranks (each rank normalises its loss by its own mask counts), then clipped and applied
identically on every rank, which keeps the replicas bit-identical without parameter broadcasts.
"""
import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def broadcast_arena(self, arena, src=0):
        dist.broadcast(arena.data, src=src, group=self.group)

    def attach(self, loop):
        """Hook a SyntheticLoop-like object (``.mac``, ``.learner``, ``.behavior``, ``.prediction``): replicas
        start from rank 0's weights and every learner all-reduces its gradient arenas before stepping."""
        arenas = [loop.mac.actor_arena, loop.mac.critic_arena]
        if getattr(loop, "behavior", None) is not None:
            arenas += [loop.behavior.enc_arena, loop.behavior.dec_arena]
            loop.behavior.dp = self
        if getattr(loop, "prediction", None) is not None:
            arenas += [loop.prediction.gat_arena, loop.prediction.dec_arena]
            loop.prediction.dp = self
        loop.learner.dp = self
        for a in arenas:
            self.broadcast_arena(a)
        return self

    def all_reduce_grads(self, *arenas):
        """Average the gradient arenas over the ranks: one collective per arena, in place."""
        works = []
        for a in arenas:
            if self.backend == "nccl":
                works.append(dist.all_reduce(a.grad, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
            else:
                works.append(dist.all_reduce(a.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if self.backend != "nccl":
            for a in arenas:
                a.grad.div_(self.world)
