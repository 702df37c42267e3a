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
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from .. import ops
from .augment import draw_intensity, draw_spatial, forward_point, to_index_map_xyz
from .distributed import broadcast_buffers, env_world, init_distributed
from .pipeline import PredictPipeline


class CachedVolumes:
    """Pre-processed volumes resident in HBM + per-class voxel index lists for crop sampling."""

    def __init__(self, files, device, num_classes: int):
        pipe = PredictPipeline(device=device, spacing=(), with_label=True)
        self.items = []
        for f in files:
            it = pipe.load(f["image"], f["label"])
            lab = it["label"][0]
            flat = lab.reshape(-1).long()
            idx = [torch.nonzero(flat == c).reshape(-1).to(torch.int32) for c in range(num_classes)]
            self.items.append({"image": it["image"], "label": it["label"], "class_idx": idx})

    def __len__(self):
        return len(self.items)


def crop_centers(rng: np.random.RandomState, item: Dict, roi, num_samples: int, num_classes: int,
                 spatial: Optional[np.ndarray] = None):
    """MONAI ``generate_label_classes_crop_centers`` + ``correct_crop_centers`` with
    ratios = [0, 1, 1, ...] (background never chosen as a centre, monai_unet.py:201).

    ``spatial`` (the pull-back map of ``augment.draw_spatial``): the centre voxel is drawn from the
    stored label's class lists and carried into the augmented volume, where the reference would
    have drawn it from the resampled label."""
    shape = list(item["label"].shape[1:])
    ratios = np.array([0.0 if c == 0 else 1.0 for c in range(num_classes)])
    counts = np.array([int(t.numel()) for t in item["class_idx"]])
    ratios = np.where(counts > 0, ratios, 0.0)
    if ratios.sum() == 0:
        ratios = (counts > 0).astype(np.float64)
    probs = ratios / ratios.sum()
    picks, pos = [], []
    for _ in range(num_samples):
        c = int(rng.choice(num_classes, p=probs))
        picks.append(c)
        pos.append(int(rng.randint(counts[c])))
    flat = torch.stack([item["class_idx"][c][p] for c, p in zip(picks, pos)]).cpu().numpy()
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


def make_batch(net, cache: CachedVolumes, vol_ids, rng) -> Dict:
    """2 volumes x num_samples crops -> {'image': [B,C,*roi] f32, 'label': [B,1,*roi] f32}."""
    roi = list(net.spatial_size)
    dev = net.device
    imgs, labs = [], []
    for vid in vol_ids:
        it = cache.items[vid]
        spatial = draw_spatial(rng, it["label"].shape[1:]) if net.augment_spatial else None
        starts = crop_centers(rng, it, roi, net.num_samples, net.num_classes, spatial)
        C = it["image"].shape[0]
        src = it["image"].permute(1, 2, 3, 0).contiguous()[None]            # NDHWC, n = 1
        out_i = torch.empty((len(starts), roi[0], roi[1], roi[2], C), dtype=torch.float32, device=dev)
        out_l = torch.empty((len(starts), roi[0], roi[1], roi[2]), dtype=torch.float32, device=dev)
        flips = [(int(rng.rand() < 0.2)) | (int(rng.rand() < 0.2) << 1) | (int(rng.rand() < 0.2) << 2)
                 for _ in starts]
        if spatial is None:
            ops.crop_patches(src, it["label"][0].contiguous(), [[0] + s for s in starts], flips, out_i, out_l)
        else:
            ops.warp_crop_patches(src, it["label"][0].contiguous(), [[0] + s for s in starts], flips,
                                  to_index_map_xyz(spatial), out_i, out_l)
        if net.augment_intensity:
            con, hist, bias, gibbs, spike = draw_intensity(rng, len(starts), roi)
            ops.intensity_augment(out_i, con, hist, bias)
            ops.kspace_augment(out_i, gibbs, spike)
        imgs.append(out_i.permute(0, 4, 1, 2, 3))
        labs.append(out_l.unsqueeze(1))
    return {"image": torch.cat(imgs).contiguous(), "label": torch.cat(labs).contiguous()}


def fit(net, output_dir: Path, max_epochs: int, early_stop_patience: int, gpu_ids,
        ckpt_name: Callable, batch_volumes: int = 2, seed: int = 0):
    if len(list(gpu_ids or [0])) > 1 and env_world()[2] == 1:
        raise RuntimeError(
            "segmantic_amd: several gpu_ids need one process per GPU: launch with "
            "`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 ...`")
    rank, local_rank, world = init_distributed()
    ids = list(gpu_ids) if gpu_ids else [0]
    dev_index = ids[local_rank % len(ids)] if world > 1 else ids[0]
    device = torch.device(f"cuda:{dev_index}")
    torch.cuda.set_device(device)
    net.to(device)
    net.train()
    net.configure_optimizers()
    if world > 1:
        net.enable_grad_sync()
        broadcast_buffers(net)
        import torch.distributed as dist
        dist.broadcast(net._engine.flat, src=0)
    print(f"[rank {rank}/{world}] caching data on {device} "
          f"({len(net.dataset.training_files())} train / {len(net.dataset.validation_files())} val volumes)")
    train_cache = CachedVolumes(net.dataset.training_files(), device, net.num_classes)
    val_cache = CachedVolumes(net.dataset.validation_files(), device, net.num_classes)
    rng = np.random.RandomState(seed + rank)      # set_determinism(seed=0), reference :229
    log_dir = Path(output_dir) / "logs"
    log_dir.mkdir(parents=True, exist_ok=True)
    top: List[tuple] = []                         # (val_dice, path), best 3
    best, bad_epochs = -math.inf, 0
    log_f = open(log_dir / "metrics.csv", "a", newline="") if rank == 0 else None
    writer = csv.writer(log_f) if log_f else None
    if writer:
        writer.writerow(["epoch", "train_loss", "val_loss", "val_dice", "lr", "epoch_seconds"])
    for epoch in range(max_epochs):
        net.current_epoch = epoch
        t0 = time.time()
        order = rng.permutation(len(train_cache))
        if world > 1:
            order = order[rank::world]
        losses = []
        for b in range(0, len(order), batch_volumes):
            batch = make_batch(net, train_cache, order[b:b + batch_volumes], rng)
            losses.append(net.training_step(batch)["loss"])
        train_loss = float(torch.stack(losses).mean().item()) if losses else float("nan")
        # ---- validation (rank 0 evaluates; every rank keeps the same stopping decision)
        if world > 1:
            broadcast_buffers(net)
        for it in val_cache.items:
            net.validation_step({"image": it["image"][None], "label": it["label"][None]})
        logs = net.on_validation_epoch_end()
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
            writer.writerow([epoch, train_loss, val_loss, val_dice, net._opt.lr, time.time() - t0])
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
    return net
