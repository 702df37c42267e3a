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


# ------------------------------------------------------------------------------------------
# the epoch loop under data parallelism (reference: Lightning DistributedSampler + DDP,
# src/segmantic/seg/monai_unet.py:278-286, 529-538)
# ------------------------------------------------------------------------------------------
def test_epoch_shard_is_a_padded_partition_with_equal_lengths():
    import numpy as np
    from segmantic_amd.seg.trainer import epoch_shard
    for n, world in ((5, 2), (9, 2), (20, 8), (3, 4), (16, 4), (1, 2)):
        for epoch in (0, 1, 7):
            shards = [epoch_shard(n, epoch, 0, r, world) for r in range(world)]
            per = -(-n // world)
            assert all(len(s) == per for s in shards), (n, world)         # same number of steps
            flat = np.concatenate(shards)
            assert set(flat.tolist()) == set(range(n))                     # everything is seen
            # the padding repeats entries of the SAME permutation (at most per*world - n of them)
            assert len(flat) - len(set(flat.tolist())) == per * world - n
        # a different permutation every epoch, the same on every rank for one epoch
        a = np.concatenate([epoch_shard(n, 0, 0, r, world) for r in range(world)])
        b = np.concatenate([epoch_shard(n, 1, 0, r, world) for r in range(world)])
        if n > 3:
            assert not np.array_equal(a, b)
    assert np.array_equal(np.sort(epoch_shard(7, 3, 0, 0, 1)), np.arange(7))   # single process: a permutation


class _StubNet:
    """Stands in for ``Net`` in ``run_epochs``: a training step is one gradient all-reduce (what
    ``GradSync`` issues), validation returns rank-dependent numbers that must be replaced by rank
    0's before any decision is taken."""

    def __init__(self, rank):
        self.rank = rank
        self.current_epoch = 0
        self._opt = None
        self.saved = []
        self.seen_lr_metric = []

    def save_checkpoint(self, path, epoch=0):
        self.saved.append(str(path))


def _fit_worker(rank, world, port, ret, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from datetime import timedelta
    from pathlib import Path
    from segmantic_amd.seg import trainer
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))
    net = _StubNet(rank)
    steps = []

    def step_fn(vol_ids, rng):
        g = torch.full((8,), float(rank + 1))
        dist.all_reduce(g)                 # a rank with an extra step would block here forever
        steps.append([int(v) for v in vol_ids])
        return float(g[0])

    epochs_seen = []

    def validate_fn():
        # each rank "measures" something different; the loop must continue with rank 0's values
        e = len(epochs_seen)
        local_dice = [0.5, 0.6, 0.55, 0.55][e] + 0.1 * rank
        local_loss = 1.0 - local_dice
        d, l = trainer.sync_from_rank0((local_dice, local_loss), "cpu")
        epochs_seen.append((d, l))
        return {"val_dice": d, "val_loss": l}

    n_steps = trainer.run_epochs(net, 5, step_fn, validate_fn, Path(tmp), max_epochs=10,
                                 early_stop_patience=2, ckpt_name=lambda o, e, l, d: Path(o) / f"e{e}.ckpt",
                                 batch_volumes=2, seed=0, rank=rank, world=world)
    # dataset split follows rank 0
    class DS:
        pass
    ds = DS()
    ds._train_files, ds._val_files, ds._test_files = [f"t{rank}"], [f"v{rank}"], []
    ds = trainer.sync_dataset(ds)
    ret[rank] = {"steps": n_steps, "epochs": epochs_seen, "vols": steps, "saved": net.saved,
                 "split": (ds._train_files, ds._val_files)}
    dist.destroy_process_group()


def test_fit_loop_world2_with_5_volumes_runs_equal_steps_and_takes_rank0_decisions(tmp_path):
    """5 training volumes on 2 ranks: round 1's loop sliced two DIFFERENT permutations
    ``[rank::world]`` (3 vs 2 volumes -> 2 vs 1 steps -> mismatched all-reduces -> hang) and let
    each rank early-stop on its own validation numbers."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_fit_worker, args=(world, port, ret, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    # ceil(5/2) = 3 volumes per rank per epoch -> 2 steps per epoch on BOTH ranks
    assert r0["steps"] == r1["steps"] and r0["steps"] == 2 * len(r0["epochs"])
    assert r0["epochs"] == r1["epochs"]                        # rank 0's metrics everywhere
    assert [round(d, 2) for d, _ in r0["epochs"]] == [0.5, 0.6, 0.55, 0.55]   # early stop after 2 bad epochs
    # one epoch: the two ranks' volumes cover 0..4 (one repeated by the padding)
    e0 = [v for s in r0["vols"][:2] for v in s] + [v for s in r1["vols"][:2] for v in s]
    assert sorted(set(e0)) == [0, 1, 2, 3, 4] and len(e0) == 6
    assert len(r0["saved"]) >= 1 and r1["saved"] == []         # only rank 0 writes checkpoints
    assert r0["split"] == r1["split"] == (["t0"], ["v0"])
