"""One-shot peer-to-peer all-reduce (iplan_amd/csrc/p2p.hip, parallel.P2PAllReduce).

CPU: the kernels on the host emulator, three "ranks" in one process (handles are plain pointers there): every rank publishes,
then every rank reduces -- sums in rank order, identical on all ranks, staging halves alternate, slices for tensors beyond the
staging capacity, the time-out path.  GPU: two PROCESSES on one device exchange real HIP IPC handles and run the collective
concurrently (the cross-device part -- system-scope fences over xGMI -- needs a multi-GPU node and is not exercised here)."""
import os

import pytest
import torch

from iplan_amd import _lib as L
from iplan_amd.parallel import P2PAllReduce


def test_p2p_allreduce_emulated_three_ranks():
    from tests.emu.emu_lib import get_emu_lib
    lib = get_emu_lib()
    world, cap = 3, 64
    ranks = [P2PAllReduce(world, r, "cpu", cap, None, lib=lib, spin_limit=10) for r in range(world)]
    handles = [p.handles for p in ranks]
    for p in ranks:
        p.connect(handles)
    g = torch.Generator().manual_seed(0)
    for it, n in enumerate((64, 8, 64, 40)):                # both staging halves, partial buffers
        xs = [torch.randn(n, generator=g) for _ in range(world)]
        want = (xs[0] + xs[1]) + xs[2]                        # rank order
        args = [p.publish(x) for p, x in zip(ranks, xs)]
        for p, a, x in zip(ranks, args, xs):
            p.reduce(a, x.device)
        for x in xs:
            assert torch.equal(x, want), it
    # beyond the capacity: slices, each a collective of its own (every rank walks the same slices in the same order)
    xs = [torch.randn(3 * cap + 8, generator=g) for _ in range(world)]
    want = (xs[0] + xs[1]) + xs[2]
    for lo in range(0, xs[0].numel(), cap):
        args = [p.publish(x[lo:lo + cap]) for p, x in zip(ranks, xs)]
        for p, a, x in zip(ranks, args, xs):
            p.reduce(a, x.device)
    assert all(torch.equal(x, want) for x in xs)
    # a rank whose peers never publish gives up after spin_limit polls and raises the error flag instead of hanging
    x = torch.ones(8)
    a = ranks[0].publish(x)
    ranks[0].reduce(a, x.device)
    assert int(ranks[0].error.item()) == 1 and torch.equal(x, torch.ones(8))
    for p in ranks:
        p.close()
    with pytest.raises(L.IplanError):
        bad = L.P2pArgs()
        lib.call("iplan_p2p_reduce", bad, 0)


def _gpu_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)

        def exchange(obj):
            got = [None] * world
            dist.all_gather_object(got, obj)
            return got

        cap = 1 << 14
        p2p = P2PAllReduce(world, rank, dev, cap, exchange, spin_limit=4_000_000)
        g = torch.Generator().manual_seed(100 + rank)
        ok = True
        for it, n in enumerate((cap, 4096, 3 * cap + 1024, 512)):
            x = torch.randn(n, generator=g)
            parts = [torch.empty(n) for _ in range(world)]
            dist.all_gather(parts, x)
            want = parts[0]
            for p in parts[1:]:
                want = want + p
            y = x.to(dev)
            p2p.all_reduce(y)
            torch.cuda.synchronize()
            ok = ok and torch.equal(y.cpu(), want) and int(p2p.error.item()) == 0
        dist.barrier()
        p2p.close()
        out.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_p2p_allreduce_two_processes_one_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 400
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)], res
