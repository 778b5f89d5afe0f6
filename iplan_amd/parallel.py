"""Data-parallel training across the GPUs of a node: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), every rank owns its shard of the parallel environments and a full replica of
all networks.

The reference has no multi-GPU path at all; this is the MI355X-native addition the
north star asks for.  Rollout inference and the three learners' forward/backward run rank-local with
no communication; the only exchange step is ONE sum all-reduce per gradient arena per optimiser step
(PPO: actor + critic arenas, 15x per train(); prediction: GAT + decoder arenas once per rollout;
behaviour: encoder + decoder arenas once per rollout) -- each arena is a single contiguous fp32
buffer holding all agents' gradients, so there are no per-tensor collectives.  Messages are 0.3-4 MB:
latency-bound on the xGMI mesh, hence as few and as large as the algorithm allows.

Semantics: an N-rank step computes exactly the gradient of ONE process that holds the union of the ranks' data.
Every loss on the path is a masked sum divided by a normaliser (sum of the mask over a window / the sampled rows /
the PPO rows; advantage mean and std), so before its backward pass each learner all-reduces those few scalars
(``all_reduce_sum``) and scales its local loss by the GLOBAL denominators; the gradient arenas are then SUMMED, clipped
and applied identically on every rank, which keeps the replicas bit-identical without parameter broadcasts.
"""
import torch.distributed as dist


class DataParallel:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def broadcast_arena(self, arena, src=0):
        dist.broadcast(arena.data, src=src, group=self.group)
        arena.version += 1

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

    def all_reduce_sum(self, tensor):
        """Sum a (small) tensor of loss normalisers over the ranks, in place; returns it."""
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        return tensor

    def all_reduce_grads(self, *arenas):
        """Sum the gradient arenas over the ranks (the local losses are already scaled by the global normalisers): one
        collective per arena, in place."""
        works = [dist.all_reduce(a.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for a in arenas]
        for w in works:
            w.wait()
