"""Training loop that stands in for ``pl.Trainer(...).fit(net)`` as configured by the reference
(``src/segmantic/seg/monai_unet.py:503-541``): epoch loop, validation every epoch, top-3
``ModelCheckpoint(monitor="val_dice", mode="max")`` with Lightning's file-name pattern,
``EarlyStopping(monitor="val_dice", patience)``, LR scheduling per validation epoch, and
one-process-per-GPU data parallelism when launched under torchrun or with several ``gpu_ids``.

Data path (reference ``:224-286``): volumes are pre-processed once and cached ON THE GPU
(``CacheDataset(cache_rate=1)``); each step draws ``batch_size=2`` volumes x ``num_samples``
crops with MONAI's ``RandCropByLabelClassesd(ratios=[0,1,1,...])`` centre rule and ``RandFlipd``
(p=0.2 per axis); the crop / flip / cast itself is one HIP kernel (``segmi_crop_patches``).
"""
from __future__ import annotations

import csv
import math
import os
import time
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import ops
from . import streams
from .augment import draw_intensity, draw_spatial, forward_point, to_index_map_xyz
from .distributed import broadcast_buffers, env_world, init_distributed
from .pipeline import PredictPipeline


class CachedVolumes:
    """Pre-processed volumes resident in HBM + per-class voxel index lists for crop sampling.

    The index lists of a volume are one device tensor (classes concatenated) with host-side
    offsets / counts, so that drawing ``num_samples`` centres is one small gather on a dedicated
    sampler stream + one pinned D2H copy: the training stream is never synchronised by the
    sampler (a ``.cpu()`` on the training stream would drain the whole step pipeline)."""

    def __init__(self, files, device, num_classes: int, spacing=(), label_nearest: bool = False):
        pipe = PredictPipeline(device=device, spacing=spacing, with_label=True, label_nearest=label_nearest)
        self.items = []
        self.device = torch.device(device)
        self._stream = streams.shared_stream(self.device, streams.AUX)
        self._pinned = torch.empty(4096, dtype=torch.int64).pin_memory()
        for f in files:
            it = pipe.load(f["image"], f["label"])
            lab = it["label"][0]
            flat = lab.reshape(-1).long()
            idx = [torch.nonzero(flat == c).reshape(-1) for c in range(num_classes)]
            counts = np.array([int(t.numel()) for t in idx], dtype=np.int64)
            offsets = np.concatenate([[0], np.cumsum(counts)[:-1]])
            self.items.append({"image": it["image"], "label": it["label"],
                               "class_all": torch.cat(idx) if idx else flat[:0],
                               "class_counts": counts, "class_offsets": offsets,
                               # NDHWC view of the image the crop kernels read (made once)
                               "image_ndhwc": it["image"].permute(1, 2, 3, 0).contiguous()[None],
                               "label_dhw": it["label"][0].contiguous()})
        torch.cuda.current_stream(self.device).synchronize()   # the cache is complete and static

    def __len__(self):
        return len(self.items)

    def lookup(self, item: Dict, positions: np.ndarray) -> np.ndarray:
        """flat voxel indices ``class_all[positions]`` -> host, without touching the training
        stream (the index lists never change after construction)."""
        n = len(positions)
        with torch.cuda.stream(self._stream):
            pos = torch.from_numpy(np.ascontiguousarray(positions, dtype=np.int64)).to(
                self.device, non_blocking=True)
            got = item["class_all"].index_select(0, pos)
            self._pinned[:n].copy_(got, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        ev.synchronize()
        return self._pinned[:n].numpy().copy()


def crop_centers(rng: np.random.RandomState, item: Dict, roi, num_samples: int, num_classes: int,
                 spatial: Optional[np.ndarray] = None, cache: Optional[CachedVolumes] = None):
    """MONAI ``generate_label_classes_crop_centers`` + ``correct_crop_centers`` with
    ratios = [0, 1, 1, ...] (background never chosen as a centre, monai_unet.py:201).

    ``spatial`` (the pull-back map of ``augment.draw_spatial``): the centre voxel is drawn from the
    stored label's class lists and carried into the augmented volume, where the reference would
    have drawn it from the resampled label."""
    shape = list(item["label"].shape[1:])
    ratios = np.array([0.0 if c == 0 else 1.0 for c in range(num_classes)])
    counts = np.asarray(item["class_counts"])
    ratios = np.where(counts > 0, ratios, 0.0)
    if ratios.sum() == 0:
        ratios = (counts > 0).astype(np.float64)
    probs = ratios / ratios.sum()
    picks, pos = [], []
    for _ in range(num_samples):
        c = int(rng.choice(num_classes, p=probs))
        picks.append(c)
        pos.append(int(rng.randint(counts[c])))
    where = np.array([item["class_offsets"][c] + p for c, p in zip(picks, pos)], dtype=np.int64)
    if cache is not None:
        flat = cache.lookup(item, where)
    else:
        flat = item["class_all"][torch.from_numpy(where).to(item["class_all"].device)].cpu().numpy()
    centers = np.stack(np.unravel_index(flat, shape), 1)
    if spatial is not None:
        centers = np.stack([np.rint(forward_point(spatial, c)).astype(np.int64) for c in centers])
    starts = []
    for ctr in centers:
        st = []
        for d in range(3):
            size, dim = roi[d], shape[d]
            if dim <= size:               # SpatialPadd: symmetric padding, crop covers everything
                st.append(-((size - dim) // 2))
                continue
            vs = size // 2
            ve = int(dim + 1 - size / 2.0)
            if vs == ve:
                ve += 1
            c = min(max(int(ctr[d]), vs), ve - 1)
            st.append(max(c - size // 2, 0))
        starts.append(st)
    return starts


def batch_buffers(net, cache: CachedVolumes, n_volumes: int) -> Dict:
    """destination of ``make_batch(..., out=)``: NDHWC image and label buffers of one batch"""
    roi = list(net.spatial_size)
    B = n_volumes * net.num_samples
    C = cache.items[0]["image"].shape[0]
    dev = cache.device
    return {"image": torch.empty((B, roi[0], roi[1], roi[2], C), dtype=torch.float32, device=dev),
            "label": torch.empty((B, roi[0], roi[1], roi[2]), dtype=torch.float32, device=dev)}


def make_batch(net, cache: CachedVolumes, vol_ids, rng, out: Optional[Dict] = None) -> Dict:
    """2 volumes x num_samples crops -> {'image': [B,C,*roi] f32, 'label': [B,1,*roi] f32}.

    ``out`` (``batch_buffers``): the crops are written straight into these buffers -- no allocation
    and no concatenation per step; the returned tensors are views of them (the image as the
    NCDHW-shaped view of the NDHWC buffer the crop kernels fill)."""
    roi = list(net.spatial_size)
    dev = net.device
    imgs, labs = [], []
    row = 0
    for vid in vol_ids:
        it = cache.items[vid]
        spatial = draw_spatial(rng, it["label"].shape[1:]) if net.augment_spatial else None
        starts = crop_centers(rng, it, roi, net.num_samples, net.num_classes, spatial, cache)
        C = it["image"].shape[0]
        src = it["image_ndhwc"]                                              # NDHWC, n = 1
        if out is not None and out["image"].shape[0] >= row + len(starts) and out["image"].shape[4] == C:
            out_i = out["image"][row:row + len(starts)]
            out_l = out["label"][row:row + len(starts)]
        else:
            out = None
            out_i = torch.empty((len(starts), roi[0], roi[1], roi[2], C), dtype=torch.float32, device=dev)
            out_l = torch.empty((len(starts), roi[0], roi[1], roi[2]), dtype=torch.float32, device=dev)
        row += len(starts)
        fp = float(getattr(net, "flip_prob", 0.2))
        flips = [(int(rng.rand() < fp)) | (int(rng.rand() < fp) << 1) | (int(rng.rand() < fp) << 2)
                 for _ in starts]
        if spatial is None:
            ops.crop_patches(src, it["label_dhw"], [[0] + s for s in starts], flips, out_i, out_l)
        else:
            ops.warp_crop_patches(src, it["label_dhw"], [[0] + s for s in starts], flips,
                                  to_index_map_xyz(spatial), out_i, out_l)
        if net.augment_intensity:
            con, hist, bias, gibbs, spike = draw_intensity(rng, len(starts), roi)
            ops.intensity_augment(out_i, con, hist, bias)
            ops.kspace_augment(out_i, gibbs, spike)
        imgs.append(out_i.permute(0, 4, 1, 2, 3))
        labs.append(out_l.unsqueeze(1))
    if out is not None:
        img = out["image"][:row].permute(0, 4, 1, 2, 3)
        return {"image": img if img.is_contiguous() else img.contiguous(), "label": out["label"][:row].unsqueeze(1)}
    return {"image": torch.cat(imgs).contiguous(), "label": torch.cat(labs).contiguous()}


class BatchPrefetcher:
    """Builds the next step's batch on a side HIP stream while the current step runs (the reference's
    DataLoader workers do the same on host cores, ``monai_unet.py:278-286``).  The draws come from the
    same ``rng`` in the same order as without it, so training is bit-identical either way.
    OFF by default (``SEGMI_PREFETCH=1`` enables it): the process already uses four streams (training,
    weight gradients, residual branch, sampler) and ROCm maps streams onto 4 hardware queues
    (``GPU_MAX_HW_QUEUES``); a fifth stream that lands on the training stream's queue serialises
    behind it and costs more than the overlap gains (5.80 vs 5.87 ms per step when it gets its own queue,
    6.4-6.5 vs 5.9 ms when it does not -- measured after other workloads had created streams).

    Two persistent buffer sets alternate (no allocation per step: tensors handed from one stream's
    allocator pool to another would keep the pool growing for several steps): set k is refilled only
    after the step that read it has been enqueued completely (``release``)."""

    def __init__(self, net, cache: CachedVolumes):
        self.net, self.cache = net, cache
        self.enabled = os.environ.get("SEGMI_PREFETCH", "0") == "1"
        self.stream = streams.shared_stream(cache.device, streams.AUX) if self.enabled else None
        self._sets: List[Optional[Dict]] = [None, None]
        self._released: List[Optional[torch.cuda.Event]] = [None, None]
        self._turn = 0

    def prepare(self, vol_ids, rng):
        """enqueue the batch of ``vol_ids``; returns a handle for ``take`` / ``release``"""
        k = self._turn
        self._turn ^= 1
        if self._sets[k] is None or self._sets[k]["image"].shape[0] < len(vol_ids) * self.net.num_samples:
            self._sets[k] = batch_buffers(self.net, self.cache, len(vol_ids))
            self._released[k] = None
        if not self.enabled:
            return (make_batch(self.net, self.cache, vol_ids, rng, out=self._sets[k]), None, k)
        # no dependency on the training stream except the buffer set's previous reader: the cache is
        # static (synchronised when it was built)
        if self._released[k] is not None:
            self.stream.wait_event(self._released[k])
        with torch.cuda.stream(self.stream):
            batch = make_batch(self.net, self.cache, vol_ids, rng, out=self._sets[k])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return (batch, ev, k)

    def take(self, handle) -> Dict:
        """the batch, ordered before everything the training stream does next"""
        batch, ev, _k = handle
        if ev is not None:
            main = torch.cuda.current_stream(self.cache.device)
            main.wait_event(ev)
            for t in batch.values():
                t.record_stream(main)       # (only matters for tensors make_batch had to allocate)
        return batch

    def release(self, handle) -> None:
        """call once the step that reads the batch has been enqueued: its buffer set may be refilled"""
        _batch, ev, k = handle
        if ev is not None:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.cache.device))
            self._released[k] = done


def epoch_shard(n: int, epoch: int, seed: int, rank: int, world: int) -> np.ndarray:
    """Volume indices of ``rank`` for ``epoch`` -- torch ``DistributedSampler`` (what Lightning
    injects for the reference's train loader, ``monai_unet.py:278-286,529-538``): ONE permutation
    per epoch drawn from a rank-independent seed, padded by wrapping to ``ceil(n / world) * world``
    entries, dealt ``rank::world``.  Every rank gets the same number of volumes, hence runs the
    same number of ``training_step`` s (= the same number of gradient all-reduces)."""
    order = np.random.RandomState((seed + epoch) % (2 ** 32)).permutation(n)
    if world <= 1:
        return order
    per_rank = -(-n // world)
    total = per_rank * world
    if total > n:
        reps = -(-total // n)
        order = np.concatenate([order] * reps)[:total]
    return order[rank:total:world]


def sync_from_rank0(values: Sequence[float], device) -> List[float]:
    """Every rank continues with rank 0's numbers (validation metrics -> LR schedule, early stop,
    checkpoint names): replicas must never take different control-flow decisions."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64,
                     device=device if dist.get_backend() == "nccl" else "cpu")
    dist.broadcast(t, src=0)
    return [float(v) for v in t.cpu()]


def sync_dataset(dataset):
    """All ranks use rank 0's train / validation / test file lists (``PairedDataSet`` shuffles with
    an unseeded RNG, so each process would otherwise draw its own split)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dataset
    box = [(dataset._train_files, dataset._val_files, dataset._test_files)] if dist.get_rank() == 0 else [None]
    dist.broadcast_object_list(box, src=0)
    dataset._train_files, dataset._val_files, dataset._test_files = box[0]
    return dataset


def run_epochs(net, n_train: int, step_fn: Callable, validate_fn: Callable, output_dir: Path,
               max_epochs: int, early_stop_patience: int, ckpt_name: Callable, batch_volumes: int,
               seed: int, rank: int, world: int, device=None, prepare_fn: Optional[Callable] = None):
    """The epoch loop of ``fit``: identical control flow on every rank.

    ``prepare_fn(volume_ids, rng) -> handle`` (optional) builds a step's batch ahead of time; the
    handle is then passed to ``step_fn(volume_ids, rng, handle)``.

    ``step_fn(volume_ids, rng) -> loss`` (scalar tensor or float) runs one ``training_step``;
    ``validate_fn() -> {"val_dice", "val_loss"}`` runs the validation epoch (its numbers are
    already rank 0's, see ``Net.on_validation_epoch_end(sync=...)``)."""
    rng = np.random.RandomState((seed + 7919 * (rank + 1)) % (2 ** 32))   # crops / flips / augmentation
    log_dir = Path(output_dir) / "logs"
    top: List[tuple] = []                         # (val_dice, path), best 3
    best, bad_epochs = -math.inf, 0
    log_f = writer = None
    if rank == 0:
        log_dir.mkdir(parents=True, exist_ok=True)
        log_f = open(log_dir / "metrics.csv", "a", newline="")
        writer = csv.writer(log_f)
        writer.writerow(["epoch", "train_loss", "val_loss", "val_dice", "lr", "epoch_seconds"])
    steps_run = 0
    for epoch in range(max_epochs):
        net.current_epoch = epoch
        t0 = time.time()
        order = epoch_shard(n_train, epoch, seed, rank, world)
        losses = []
        groups = [order[b:b + batch_volumes] for b in range(0, len(order), batch_volumes)]
        if prepare_fn is None:
            for ids in groups:
                losses.append(step_fn(ids, rng))
                steps_run += 1
        else:
            # step i is enqueued first, then the batch of step i + 1 is built beside it
            pending = prepare_fn(groups[0], rng) if groups else None
            for i, ids in enumerate(groups):
                losses.append(step_fn(ids, rng, pending))
                pending = prepare_fn(groups[i + 1], rng) if i + 1 < len(groups) else None
                steps_run += 1
        if losses and torch.is_tensor(losses[0]):
            train_loss = float(torch.stack([l.reshape(()) for l in losses]).mean().item())
        else:
            train_loss = float(np.mean(losses)) if losses else float("nan")
        logs = validate_fn()
        val_dice, val_loss = logs["val_dice"], logs["val_loss"]
        if rank == 0:
            path = ckpt_name(output_dir, epoch, val_loss, val_dice)
            if not math.isnan(val_dice) and (len(top) < 3 or val_dice > min(t[0] for t in top)):
                net.save_checkpoint(path, epoch=epoch)
                top.append((val_dice, path))
                top.sort(key=lambda t: -t[0])
                for _, p in top[3:]:
                    if Path(p).exists():
                        Path(p).unlink()
                top = top[:3]
            writer.writerow([epoch, train_loss, val_loss, val_dice, net._opt.lr if net._opt else "",
                             time.time() - t0])
            log_f.flush()
        if not math.isfinite(val_dice):           # EarlyStopping(check_finite=True)
            print("val_dice is not finite: stopping")
            break
        if val_dice > best:
            best, bad_epochs = val_dice, 0
        else:
            bad_epochs += 1
            if bad_epochs >= early_stop_patience:
                print(f"early stopping at epoch {epoch} (no val_dice improvement for {bad_epochs} epochs)")
                break
    if log_f:
        log_f.close()
    return steps_run


def fit(net, output_dir: Path, max_epochs: int, early_stop_patience: int, gpu_ids,
        ckpt_name: Callable, batch_volumes: int = 2, seed: int = 0):
    if len(list(gpu_ids or [0])) > 1 and env_world()[2] == 1:
        raise RuntimeError(
            "segmantic_amd: several gpu_ids need one process per GPU: call monai_unet.train() / the "
            "segmantic-unet CLI (they start the ranks themselves, seg/launch.py) or launch with "
            "`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 ...`")
    ids = list(gpu_ids) if gpu_ids else [0]
    _r, local_rank, world = env_world()
    dev_index = ids[local_rank % len(ids)] if world > 1 else ids[0]
    rank, local_rank, world = init_distributed(device_index=dev_index)
    device = torch.device(f"cuda:{dev_index}")
    torch.cuda.set_device(device)
    net.to(device)
    net.train()
    net.configure_optimizers()
    if world > 1:
        net.dataset = sync_dataset(net.dataset)
        net.enable_grad_sync()
        broadcast_buffers(net)
        import torch.distributed as dist
        dist.broadcast(net._engine.flat, src=0)
    print(f"[rank {rank}/{world}] caching data on {device} "
          f"({len(net.dataset.training_files())} train / {len(net.dataset.validation_files())} val volumes)")
    spacing = list(getattr(net, "train_spacing", []) or [])
    ln = bool(getattr(net, "train_spacing_label_nearest", False))
    train_cache = CachedVolumes(net.dataset.training_files(), device, net.num_classes, spacing, ln)
    val_cache = CachedVolumes(net.dataset.validation_files(), device, net.num_classes, spacing, ln)

    prefetch = BatchPrefetcher(net, train_cache)

    def step_fn(vol_ids, rng, handle):            # set_determinism(seed=0), reference :229
        loss = net.training_step(prefetch.take(handle))["loss"]
        prefetch.release(handle)
        return loss

    def validate_fn():
        if world > 1:
            broadcast_buffers(net)
        for it in val_cache.items:
            net.validation_step({"image": it["image"][None], "label": it["label"][None]})
        from .inferers import release_workspaces
        release_workspaces()      # the validation volumes' prediction cache goes back before the next training epoch
        return net.on_validation_epoch_end(sync=lambda *v: sync_from_rank0(v, device))

    run_epochs(net, len(train_cache), step_fn, validate_fn, output_dir, max_epochs,
               early_stop_patience, ckpt_name, batch_volumes, seed, rank, world, device,
               prepare_fn=prefetch.prepare)
    return net
