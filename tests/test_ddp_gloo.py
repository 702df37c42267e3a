"""world_size-2 CPU (gloo) test of the data-parallel plumbing: the bucketed, backward-overlapped
all-reduce of the flat gradient arena (``GradSync``) must equal a plain all-reduce whatever the
bucket size and the order / granularity of the ``ready()`` calls, and buffers follow rank 0."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from segmantic_amd.seg.distributed import GradSync, broadcast_buffers, init_distributed
    r, lr, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    n = 100_003
    g = torch.Generator().manual_seed(100 + rank)
    ok = True
    for bucket_bytes, cuts in ((4 << 10, [90_000, 50_001, 50_000, 12, 0]), (1 << 20, [60_000, 0]),
                               (64, [99_999, 3, 0])):
        grad = torch.randn(n, generator=g)
        ref = grad.clone()
        dist.all_reduce(ref)
        gs = GradSync(grad, bucket_bytes=bucket_bytes)
        assert gs.world == world and abs(gs.grad_scale - 1.0 / world) < 1e-12
        gs.start()
        for lo in cuts:          # gradients at offsets >= lo are final (backward order)
            gs.ready(lo)
        gs.finish()
        ok = ok and bool(torch.allclose(grad, ref, rtol=0, atol=0))
    # buffers follow rank 0
    m = torch.nn.BatchNorm1d(4)
    m.running_mean.fill_(float(rank + 1))
    broadcast_buffers(m)
    ok = ok and bool((m.running_mean == 1.0).all())
    ret[rank] = ok
    dist.destroy_process_group()


def test_gradsync_bucketed_allreduce_gloo_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _slab_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from segmantic_amd.seg.distributed import init_distributed
    from segmantic_amd.seg.inferers import gather_label_slabs, z_slabs
    init_distributed(backend="gloo")
    depth = 5                                     # uneven: slabs (0,3) and (3,5)
    full = (torch.arange(depth * 4 * 3) % 251).to(torch.uint8).reshape(1, 1, depth, 4, 3)
    a, b = z_slabs(depth, world)[rank]
    got = gather_label_slabs(full[:, :, a:b].clone(), depth, rank, world)
    ret[rank] = bool(torch.equal(got, full))
    dist.destroy_process_group()


def test_sharded_inference_label_slab_gather_gloo_world2():
    """Multi-GPU form of one sliding-window volume: ranks own z-slabs, only label slabs travel."""
    from segmantic_amd.seg.inferers import z_slabs
    assert z_slabs(512, 8) == [(64 * r, 64 * r + 64) for r in range(8)]
    assert z_slabs(5, 2) == [(0, 3), (3, 5)] and z_slabs(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_slab_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
