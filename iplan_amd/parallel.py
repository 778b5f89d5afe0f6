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
import ctypes as C
import os

import torch
import torch.distributed as dist


class P2PAllReduce:
    """One-shot peer-to-peer sum all-reduce over xGMI (include/iplan_hip.h, ``IplanP2pArgs``; opt-in: IPLAN_P2P_ALLREDUCE=1).

    Every rank owns a double-buffered staging area and a flag array that its peers map through HIP IPC handles (exchanged
    once through ``exchange``, normally ``dist.all_gather_object``).  ``all_reduce(tensor)`` = publish (copy into the staging
    half, one flag store into every rank's flag array) + reduce (wait for all flags, sum all staging halves in rank order,
    in place): two launches, no host synchronisation, replicas bit-identical.  Tensors larger than the staging capacity go
    through in slices.  ``exchange(obj) -> [obj of rank 0, ...]``.

    STREAMS.  The double-buffer argument (a rank can publish ``seq + 2`` only after it finished ``seq + 1``) holds only if a
    rank's collectives execute in the order their sequence numbers were handed out.  The training loop issues them from
    several producer streams (prediction / PPO learner streams, main, the deferred decoder update's side stream), so on CUDA
    every collective runs on ONE dedicated per-rank communication stream, the way RCCL does it: the comm stream waits for an
    event of the producer stream, publish and reduce are enqueued there in host order (= ``seq`` order, the same on every
    rank), and the producer stream then waits for the collective's completion event.  No host synchronisation either way."""

    def __init__(self, world, rank, device, capacity_floats, exchange, lib=None, spin_limit=0):
        from . import _lib as L
        assert 1 <= world <= L.P2P_MAX_RANKS and capacity_floats % 4 == 0
        self.L, self.lib = L, (lib or L.get_lib())
        self.world, self.rank, self.device, self.capacity = world, rank, torch.device(device), int(capacity_floats)
        self.spin_limit, self.seq = int(spin_limit), 0
        self._own, self._opened = [], []
        with self._dev():
            self._stage, self._flags = self._alloc(2 * self.capacity * 4), self._alloc(L.P2P_MAX_RANKS * 4)
        self.handles = (bytes(self._export(self._stage).bytes), bytes(self._export(self._flags).bytes))
        self.error = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.stage = self.flags = None
        self.comm = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        if exchange is not None:
            self.connect(exchange(self.handles))

    def connect(self, handles):
        """``handles[p]`` = rank p's ``.handles``: map every peer's staging area and flag array"""
        self.stage, self.flags = [None] * self.world, [None] * self.world
        with self._dev():
            for p, (hs, hf) in enumerate(handles):
                self.stage[p], self.flags[p] = (self._stage, self._flags) if p == self.rank else (self._open(hs), self._open(hf))

    def _dev(self):
        import contextlib
        return torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()

    def _check(self, rc, what):
        if rc != 0:
            raise self.L.IplanError(f"{what} failed ({rc}): {self.lib.c.iplan_last_error().decode()}")

    def _alloc(self, nbytes):
        p = C.c_void_p()
        self._check(self.lib.c.iplan_p2p_alloc(nbytes, C.byref(p)), "iplan_p2p_alloc")
        self._own.append(p.value)
        return p.value

    def _export(self, ptr):
        h = self.L.IpcHandle()
        self._check(self.lib.c.iplan_p2p_export(ptr, C.byref(h)), "iplan_p2p_export")
        return h

    def _open(self, raw):
        h = self.L.IpcHandle()
        C.memmove(h.bytes, raw, 64)
        p = C.c_void_p()
        self._check(self.lib.c.iplan_p2p_open(C.byref(h), C.byref(p)), "iplan_p2p_open")
        self._opened.append(p.value)
        return p.value

    def _args(self, ptr, count):
        a = self.L.P2pArgs()
        a.world, a.rank, a.count, a.data, a.capacity = self.world, self.rank, count, ptr, self.capacity
        for p in range(self.world):
            a.stage[p], a.flags[p] = self.stage[p], self.flags[p]
        a.seq, a.error, a.spin_limit = self.seq, self.error.data_ptr(), self.spin_limit
        return a

    def publish(self, tensor):
        """first half of ``all_reduce`` for one slice (<= capacity floats): returns the argument block for ``reduce``"""
        assert tensor.dtype == torch.float32 and tensor.is_contiguous() and tensor.numel() <= self.capacity
        n = tensor.numel()
        assert n % 4 == 0, "pad gradient arenas to a multiple of 4 floats"
        self.seq += 1
        a = self._args(tensor.data_ptr(), n)
        self.lib.call("iplan_p2p_publish", a, self.L.current_stream(tensor.device))
        return a

    def reduce(self, a, device):
        self.lib.call("iplan_p2p_reduce", a, self.L.current_stream(device))

    def all_reduce(self, tensor):
        flat = tensor.view(-1)
        if self.comm is None:                                 # host emulation: one implicit stream
            for lo in range(0, flat.numel(), self.capacity):
                part = flat[lo:lo + self.capacity]
                self.reduce(self.publish(part), part.device)
            return tensor
        producer = torch.cuda.current_stream(self.device)
        self.comm.wait_stream(producer)                       # (an event of the producer: the gradients are complete)
        with torch.cuda.stream(self.comm):
            for lo in range(0, flat.numel(), self.capacity):
                part = flat[lo:lo + self.capacity]
                self.reduce(self.publish(part), part.device)
        tensor.record_stream(self.comm)
        producer.wait_stream(self.comm)
        return tensor

    def check_error(self):
        """Raise if a reduce gave up waiting for a peer (``spin_limit`` > 0): the tensor it was called on is then NOT reduced
        (or only partly), and carrying on would let the replicas diverge silently.  Synchronises the device."""
        if int(self.error.item()) != 0:
            raise self.L.IplanError(f"P2P all-reduce: rank {self.rank} timed out waiting for a peer's flag (seq <= {self.seq}); "
                                    "gradients were not reduced")

    def close(self):
        if self.device.type == "cuda" and (self._opened or self._own):
            torch.cuda.synchronize(self.device)
        for p in self._opened:
            self.lib.c.iplan_p2p_close(p)
        for p in self._own:
            self.lib.c.iplan_p2p_free(p)
        self._opened, self._own = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Bucket:
    """a gradient buffer with the one attribute all_reduce_grads reads of an arena"""

    def __init__(self, grad):
        self.grad = grad


def init_from_env(device=None, backend=None):
    """One process per GPU, started by ``python -m torch.distributed.run --nproc-per-node N ...`` (or ``bench.py --gpus N``): read
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, bind this process to ITS GPU, create the process group
    (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU devices) and return the ``DataParallel`` handle of the default group.
    ``device``: None = ``cuda:LOCAL_RANK`` when a GPU is visible, else "cpu"."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if device is None:
        device = torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    if device.type == "cuda":
        if device.index is None:
            device = torch.device("cuda", local_rank)
        if device.index >= torch.cuda.device_count():
            raise RuntimeError(f"rank {os.environ.get('RANK', '0')} (LOCAL_RANK {local_rank}) has no GPU of its own: "
                               f"{torch.cuda.device_count()} visible")
        torch.cuda.set_device(device)
    if not dist.is_initialized():
        backend = backend or ("nccl" if device.type == "cuda" else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device.type == "cuda") else {}
        dist.init_process_group(backend, **kw)
    dp = DataParallel(dist.group.WORLD)
    dp.device = device
    return dp


def shard_args(args, world, rank, scaling="strong"):
    """The ``args`` namespace of ONE rank of a ``world``-rank data-parallel run, derived from the single-process namespace
    (config/default.yaml values: batch_size_run 32, buffer_size 256, batch_size 255).  Returns a shallow copy; ``args`` is not
    modified.

    * ``scaling="weak"``: every rank runs the configured job (its own ``batch_size_run`` envs, its own ``buffer_size``-episode PPO
      buffer); the union is ``world`` times the configured batch.  Nothing to change: the copy only records the mode.
    * ``scaling="strong"`` (BASELINE config 4): the configured job is ONE job sharded over the ranks -- ``batch_size_run``,
      ``buffer_size`` and ``batch_size`` are the union's and must be divisible by ``world``; rank r owns envs
      ``[r E / world, (r + 1) E / world)`` and those episodes of the global buffer.  The reference trains on the FIRST
      ``batch_size`` (= buffer_size - 1) episodes of the buffer (learners/ippo_learner.py:370-372): the episodes it drops are
      the LAST ones of the union, i.e. the last rank's.  ``dp_global_rows`` / ``dp_global_count`` carry the union's PPO row /
      stored-entry counts for the advantage statistics and the entropy mean (``DataParallel.attach`` hands them to the learner)."""
    import copy
    out = copy.copy(args)
    out.dp_world, out.dp_rank, out.dp_scaling = world, rank, scaling
    if scaling == "weak" or world == 1:
        out.dp_global_rows = out.dp_global_count = None
        return out
    if scaling != "strong":
        raise ValueError(f"scaling must be 'weak' or 'strong', not {scaling!r}")
    E, B = args.batch_size_run, args.buffer_size
    if E % world or B % world:
        raise ValueError(f"strong scaling shards batch_size_run = {E} and buffer_size = {B} over {world} ranks: both must be divisible")
    drop = B - args.batch_size                              # episodes of the union's buffer train() does not use (reference: 1)
    if not 0 <= drop < B // world:
        raise ValueError(f"batch_size = {args.batch_size} must leave the last rank at least one of its {B // world} episodes")
    T = args.episode_length if getattr(args, "env", None) == "MPE" and hasattr(args, "episode_length") else args.episode_limit
    out.batch_size_run, out.buffer_size = E // world, B // world
    out.batch_size = B // world - (drop if rank == world - 1 else 0)
    out.dp_global_rows, out.dp_global_count = (B - drop) * T, B * T
    return out


class DataParallel:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.p2p = None                                      # P2PAllReduce, built on first use
        self.use_p2p = bool(os.environ.get("IPLAN_P2P_ALLREDUCE"))   # opt-in: one-shot peer-to-peer all-reduce for the gradient arenas

    def _via_host(self, tensor):
        """gloo process groups (CPU tests; the two-processes-on-one-GPU test, where RCCL refuses the duplicate device) move
        CUDA tensors through the host"""
        return self.backend == "gloo" and tensor.is_cuda

    def broadcast_arena(self, arena, src=0):
        if self._via_host(arena.data):
            h = arena.data.cpu()
            dist.broadcast(h, src=src, group=self.group)
            arena.data.copy_(h)
        else:
            dist.broadcast(arena.data, src=src, group=self.group)
        arena.version += 1

    def close(self):
        if self.p2p is not None:
            self.p2p.close()
            self.p2p = None

    def attach(self, loop=None, *, mac=None, learner=None, behavior=None, prediction=None, runner=None):
        """Hook the objects of one training process: replicas start from rank 0's weights, every learner all-reduces its loss
        normalisers and its gradient arenas before stepping, and the runner counts environment steps over ALL ranks.

        Two call forms:
        * ``attach(loop)`` -- a ``harness.SyntheticLoop``-like object with ``.mac`` / ``.learner`` / ``.behavior`` / ``.prediction``;
        * ``attach(mac=, learner=, behavior=, prediction=, runner=)`` -- the four objects ``run_ippo.run_sequential`` builds
          (run_ippo.py:194-222: ``DcntrlMAC``, ``IPPOLearner``, ``Behavior_policy`` or None, ``Prediction_policy`` or None) and its
          ``ParallelRunner``; call it once, after ``runner.setup(...)`` / ``learner.cuda()`` / ``load_models`` and before the first
          ``runner.run()``.

        If the learner's ``args`` came from ``shard_args`` in strong mode, the union's row counts it carries
        (``dp_global_rows`` / ``dp_global_count``) are handed to the learner here."""
        if loop is not None:
            mac = mac if mac is not None else loop.mac
            learner = learner if learner is not None else loop.learner
            behavior = behavior if behavior is not None else getattr(loop, "behavior", None)
            prediction = prediction if prediction is not None else getattr(loop, "prediction", None)
        if mac is None and learner is not None:
            mac = learner.mac
        if mac is None or learner is None:
            raise ValueError("DataParallel.attach needs the controller and the PPO learner (attach(loop) or attach(mac=, learner=, ...))")
        arenas = [mac.actor_arena, mac.critic_arena]
        if behavior is not None:
            if hasattr(behavior, "join_decoder"):
                behavior.join_decoder()                      # (a deferred decoder update still in flight would race the broadcast)
            arenas += [behavior.enc_arena, behavior.dec_arena]
            behavior.dp = self
        if prediction is not None:
            arenas += [prediction.gat_arena, prediction.dec_arena]
            prediction.dp = self
        learner.dp = self
        largs = getattr(learner, "args", None)
        if getattr(largs, "dp_global_rows", None) is not None:
            learner.dp_global_rows, learner.dp_global_count = largs.dp_global_rows, largs.dp_global_count
        if runner is not None:
            runner.dp = self
        for a in arenas:
            self.broadcast_arena(a)
        return self

    def broadcast_tensor(self, tensor, src=0):
        """rank ``src``'s values into every rank's ``tensor`` (same shape / dtype everywhere), in place; returns it."""
        if self._via_host(tensor):
            h = tensor.cpu()
            dist.broadcast(h, src=src, group=self.group)
            tensor.copy_(h)
        else:
            dist.broadcast(tensor, src=src, group=self.group)
        return tensor

    def all_reduce_sum(self, tensor):
        """Sum a (small) tensor of loss normalisers over the ranks, in place; returns it."""
        if self._via_host(tensor):
            h = tensor.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            tensor.copy_(h)
            return tensor
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        return tensor

    @staticmethod
    def _buckets(arenas):
        """the tensors to exchange for ``arenas``: arenas whose gradients live in one contiguous buffer
        (ParamArena.colocate_grads) and are ALL part of this call go as that one buffer -- one collective instead of one each"""
        out, seen = [], set()
        for a in arenas:
            grp = getattr(a, "_grad_group", None)
            if grp is not None and all(any(m is b for b in arenas) for m in grp[1]):
                if id(grp[0]) not in seen:
                    seen.add(id(grp[0]))
                    out.append(grp[0])
            else:
                out.append(a.grad)
        return out

    def all_reduce_grads(self, *arenas):
        """Sum the gradient arenas over the ranks (the local losses are already scaled by the global normalisers), in place:
        one collective per gradient BUFFER (co-located arenas of one call share one, see ``_buckets``)."""
        arenas = [_Bucket(t) for t in self._buckets(arenas)]
        if self.use_p2p and self.world > 1 and arenas and arenas[0].grad.is_cuda:
            if self.p2p is None:
                def exchange(obj):
                    out = [None] * self.world
                    dist.all_gather_object(out, obj, group=self.group)
                    return out
                cap = int(os.environ.get("IPLAN_P2P_CAPACITY_FLOATS", str(2 << 20)))       # 8 MB halves: the largest arena is 4 MB
                self.p2p = P2PAllReduce(self.world, self.rank, arenas[0].grad.device, cap, exchange,
                                        spin_limit=int(os.environ.get("IPLAN_P2P_SPIN_LIMIT", "0")))
            for a in arenas:
                if a.grad.numel() % 4 == 0 and a.grad.data_ptr() % 16 == 0:
                    self.p2p.all_reduce(a.grad)
                else:
                    self.all_reduce_sum(a.grad)
            if self.p2p.spin_limit > 0:                          # the time-out mode: never carry on with unreduced gradients
                self.p2p.check_error()
            return
        if arenas and self._via_host(arenas[0].grad):
            for a in arenas:
                self.all_reduce_sum(a.grad)
            return
        works = [dist.all_reduce(a.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for a in arenas]
        for w in works:
            w.wait()
