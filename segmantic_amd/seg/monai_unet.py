"""Drop-in for ``segmantic.seg.monai_unet`` (reference ``src/segmantic/seg/monai_unet.py``).

Same public names and argument meaning -- ``Net``, ``train``, ``predict`` -- but nothing below
this module is MONAI / Lightning: the network runs on the hand-written HIP kernels of
``libsegmi.so`` through ``UNetEngine``; the training loop, checkpoint format, early stopping
and top-k checkpointing of Lightning (``:503-541``) are re-stated in ~100 lines of host code.

``mixed_precision=True`` selects bf16 storage with f32 accumulation (the reference's
``precision=16`` AMP, ``:533``); ``False`` selects the exact-f32 MFMA path used for parity.
"""
import json
import os
import re
from collections.abc import Sequence
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .. import ops
from ..image.labels import load_decathlon_tissuelist, load_tissue_list
from .distributed import (GradSync, broadcast_buffers, env_world, init_distributed,
                          rank_device_index)
from .inferers import SlidingWindowInferer, sliding_window_inference
from .losses import (ConfusionMatrixMetric, DiceLoss, DiceMetric, as_ndhwc, dice_backward,
                     dice_forward)
from .optim import make_optimizer, make_scheduler
from .unet import UNetEngine, UNetParams
from .utils import make_device


class AttributeDict(dict):
    """``hparams`` container (attribute + item access), as Lightning's ``save_hyperparameters``."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _UNetFn(torch.autograd.Function):
    """Bridges the engine's explicit backward into autograd for external training loops."""

    @staticmethod
    def forward(ctx, x, anchor, net):
        ctx.net = net
        return net._to_reference_layout(net._engine_for(x).forward(x, train=True), x)

    @staticmethod
    def backward(ctx, g):
        eng = ctx.net._engine
        if g.dim() == 4:                       # 2-D network called with [N, C, H, W]
            g = g.unsqueeze(2)
        gd = as_ndhwc(g)
        if gd.dtype != eng.dtype:
            gd = gd.to(eng.dtype)
        eng.backward(gd)      # writes p.grad (views of the flat gradient arena) directly
        return None, None, None


class Net(torch.nn.Module):
    """``segmantic.seg.monai_unet.Net`` (reference ``:75-397``) on MI355X kernels."""

    cache_rate: float = 1.0
    config_preprocessing: dict = {}
    config_augmentation: dict = {}
    augment_intensity: bool = False
    augment_spatial: bool = False
    num_samples: int = 4
    flip_prob: float = 0.2            # RandFlipd(prob=0.2) per axis, reference :214-217
    train_spacing: list = []          # Spacingd(pixdim) of a configured training pre-processing
    train_spacing_label_nearest: bool = False   # ... with mode=[bilinear, nearest] (label not interpolated)
    optimizer: dict = {"optimizer": "Adam", "lr": 1e-4, "momentum": 0.9, "epsilon": 1e-8,
                       "amsgrad": False, "weight_decouple": False}
    lr_scheduling: dict = {"scheduler": "Constant", "factor": 0.5, "patience": 10, "T_0": 50,
                           "T_multi": 1}

    def __init__(self, num_classes: int, num_channels: int = 1, spatial_dims: int = 3,
                 spatial_size: Sequence[int] = None,
                 channels: tuple = (16, 32, 64, 128, 256), strides: tuple = (2, 2, 2, 2),
                 dropout: float = 0.0, act: str = "PRELU"):
        super().__init__()
        self.hparams = AttributeDict(num_classes=num_classes, num_channels=num_channels,
                                     spatial_dims=spatial_dims, spatial_size=spatial_size,
                                     channels=channels, strides=strides, dropout=dropout, act=act)
        self._model = UNetParams(spatial_dims=spatial_dims, in_channels=num_channels,
                                 out_channels=num_classes, channels=channels, strides=strides,
                                 num_res_units=2, act=act, dropout=dropout)
        self.spatial_size = list(spatial_size) if spatial_size else [96] * 3
        self.loss_function = DiceLoss(to_onehot_y=True, softmax=True)
        self.dice_metric = DiceMetric(num_classes, include_background=False)
        self.best_val_dice = 0.0
        self.best_val_epoch = 0.0
        self.current_epoch = 0
        self.validation_step_outputs: List[dict] = []
        self.mixed_precision = False
        self._engine: Optional[UNetEngine] = None
        self._anchor = torch.zeros((), requires_grad=True)
        self._opt = None
        self._sched = None
        self._gsync: Optional[GradSync] = None
        self.dataset = None

    # ------------------------------------------------------------------ properties
    @property
    def num_classes(self):
        return self._model.out_channels

    @property
    def spatial_dims(self):
        return self._model.dimensions

    @property
    def device(self):
        return next(self._model.parameters()).device

    @property
    def compute_dtype(self):
        return torch.bfloat16 if self.mixed_precision else torch.float32

    # ------------------------------------------------------------------ engine
    def _engine_for(self, x: Optional[torch.Tensor] = None) -> UNetEngine:
        dev = self.device
        if dev.type != "cuda":
            if x is not None and x.is_cuda:
                self.to(x.device)
                dev = x.device
            else:
                raise RuntimeError(
                    "segmantic_amd.Net computes on an MI355X only: move the module to a cuda "
                    "device (net.to('cuda:0')).  There is no CPU execution path.")
        if self._engine is None or self._engine.device != dev or self._engine.dtype != self.compute_dtype:
            with torch.cuda.device(dev):
                self._engine = UNetEngine(self._model, dev, self.compute_dtype)
            self._opt = None
        return self._engine

    def _apply(self, fn, *a, **k):  # .to() / .cuda(): parameters leave the arena -> rebuild lazily
        out = super()._apply(fn, *a, **k)
        self._engine = None
        self._opt = None
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        if self._engine is not None:
            self._engine.sync_weights()
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        if self._engine is not None:
            self._engine.bump()
        return out

    def _weights_final(self):
        """parameters alias the engine's arena; the tail of the last training step (the carried layers' update)
        runs on the weight-gradient stream: the READER's current stream waits for it before it touches them"""
        if self._engine is not None:
            with torch.cuda.device(self._engine.device):
                self._engine.sync_weights()

    def state_dict(self, *a, **k):
        self._weights_final()
        ops.check_fused_timeouts("state_dict")
        return super().state_dict(*a, **k)

    def named_parameters(self, *a, **k):
        # every reader of the weights on the caller's stream -- ``parameters()`` goes through here: gradient
        # clipping, EMA, logging -- is ordered behind the carried update of the last ``training_step``
        # (round 3 covered state_dict / load_state_dict / eval forwards only)
        self._weights_final()
        return super().named_parameters(*a, **k)

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[B, C, D, H, W] float32 -> logits [B, K, D, H, W] (NDHWC storage, compute dtype)."""
        eng = self._engine_for(x)
        with torch.cuda.device(eng.device):
            if self.training and torch.is_grad_enabled():
                return _UNetFn.apply(x, self._anchor, self)
            return self._to_reference_layout(eng.forward(x, train=self.training), x)

    @staticmethod
    def _to_reference_layout(logits_ndhwc: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """NDHWC storage -> logical [N, K, D, H, W]; [N, K, H, W] when a 2-D network was given
        a 4-D input."""
        out = logits_ndhwc.permute(0, 4, 1, 2, 3)
        return out.squeeze(2) if x.dim() == 4 else out

    def forward_into(self, x: torch.Tensor, out_ndhwc: torch.Tensor, lane: int = 0) -> bool:
        """Inference forward straight into a caller-owned NDHWC [B, D, H, W, K] buffer of the
        compute dtype (the sliding-window prediction cache).  Returns False when the module is in
        training mode or the buffer does not fit, so the caller falls back to ``forward``."""
        eng = self._engine_for(x)
        if (self.training or out_ndhwc.dtype != eng.dtype or not out_ndhwc.is_contiguous()
                or eng.kpad != eng.net.out_channels):
            return False
        if isinstance(x, ops.WindowBatch) and not eng.window_views_ok(x.dtype):
            return False
        with torch.cuda.device(eng.device):
            eng.forward(x, train=False, out=out_ndhwc, lane=lane)
        return True

    def cache_spec(self):
        """(classes, dtype) of what ``forward_into`` writes per voxel, or None where it would refuse: lets the
        sliding-window driver allocate its prediction cache before the first window group"""
        if self.training or self.device.type != "cuda":
            return None
        eng = self._engine_for()
        if eng.kpad != eng.net.out_channels:
            return None
        return eng.net.out_channels, eng.dtype

    def eval_state(self):
        """(engine identity, weights version): the sliding-window driver runs the FIRST window group after a change
        of it on the caller's stream before it forks its lanes -- that forward builds the folded / merged weight packs
        every lane then reads"""
        eng = self._engine_for()
        return id(eng), eng.weights_version

    def window_views_ok(self, dtype) -> bool:
        """sliding-window driver: may it hand ``forward_into`` an ``ops.WindowBatch`` (windows read in place)?"""
        if self.training or self.device.type != "cuda":
            return False
        return self._engine_for().window_views_ok(dtype)

    # ------------------------------------------------------------------ optimisers
    def configure_optimizers(self):
        eng = self._engine_for()
        self._opt = make_optimizer(self.optimizer, eng.flat, eng.flat_grad)
        self._sched = make_scheduler(self.lr_scheduling, self._opt)
        return [self._opt], [self._sched]

    def optimizers(self):
        if self._opt is None:
            self.configure_optimizers()
        return self._opt

    def lr_schedulers(self):
        if self._sched is None:
            self.configure_optimizers()
        return self._sched

    def enable_grad_sync(self, bucket_bytes: int = 4 << 20):
        """Data-parallel training: all-reduce the gradient arena overlapped with backward."""
        eng = self._engine_for()
        self._gsync = GradSync(eng.flat_grad, bucket_bytes)
        eng.grad_hook = self._gsync.ready
        return self._gsync

    # ------------------------------------------------------------------ steps
    def training_step(self, batch, batch_idx=0):
        """reference ``:339-348``: forward -> zero_grad -> Dice -> backward -> optimizer.step,
        as explicit kernel sequences (no autograd graph)."""
        images, labels = batch["image"], batch["label"]
        eng = self._engine_for(images)
        # a one-launch BatchNorm backward of an earlier step that gave up its bounded wait wrote NaN gradients:
        # stop here, loudly (host-visible counter, no device sync; csrc/norm_act.hip)
        ops.check_fused_timeouts("training_step")
        with torch.cuda.device(eng.device):
            opt = self.optimizers()
            logits = eng.forward(images, train=True)
            st = self.loss_function._state
            loss = dice_forward(st, logits, labels, self.loss_function.smooth_nr,
                                self.loss_function.smooth_dr)
            dlogits = dice_backward(st, logits, 1.0, eng.dlogits_buffer(logits),
                                    bias_grad=eng.top_bias_grad())
            if self._gsync is not None:
                self._gsync.start()
            carried = eng.backward(dlogits, top_bias_done=True, carry=self._gsync is None)
            scale = 1.0
            if self._gsync is not None:
                self._gsync.finish()
                scale = self._gsync.grad_scale
            if carried:
                # the weight gradients of the two full-resolution decoder convolutions are still running on
                # the weight-gradient stream (they overlap the NEXT step's forward, UNetEngine.carry_top_wgrad):
                # update the rest of the arena here, the suffix behind them on their stream
                opt.step(scale, 0, eng.carry_lo)
                eng.finish_carried(lambda: opt.step(scale, eng.carry_lo, None, advance=False),
                                   eng.weights_version + 1)
            else:
                opt.step(scale)
            eng.bump()
        return {"loss": loss}

    def validation_step(self, batch, batch_idx=0):
        """reference ``:350-363``: sliding window roi 160^d, sw_batch 4, Dice loss + metric."""
        images, labels = batch["image"], batch["label"]
        roi_size = tuple(160 for _ in range(self.spatial_dims))
        was = self.training
        self.eval()
        with torch.no_grad():
            res = sliding_window_inference(images.to(self.device), roi_size, 4, self.forward,
                                           return_labels=True)
            loss = self.loss_function(res.logits, labels)
            d = self.dice_metric(res.labels, labels.to(self.device).long())
        self.train(was)
        out = {"val_loss": loss, "val_number": images.shape[0], "dice": d}
        self.validation_step_outputs.append(out)
        return out

    def on_validation_epoch_end(self, sync=None):
        """reference ``:365-397``.  ``sync(val_dice, val_loss) -> (val_dice, val_loss)``: data-parallel
        runs pass a broadcast from rank 0 here, so that the LR scheduler, ``best_val_dice`` and the
        caller's early-stop / checkpoint logic see the same numbers on every rank."""
        val_loss, num_items = 0.0, 0
        for o in self.validation_step_outputs:
            val_loss += float(o["val_loss"].sum().item())
            num_items += o["val_number"]
        mean_val_dice = float(self.dice_metric.aggregate().item())
        self.dice_metric.reset()
        mean_val_loss = val_loss / max(num_items, 1)
        self.validation_step_outputs.clear()
        ops.check_fused_timeouts("validation epoch")       # the .item() above has synchronised: the count is current
        if sync is not None:
            mean_val_dice, mean_val_loss = sync(mean_val_dice, mean_val_loss)
        sched = self.lr_schedulers()
        sched.step(mean_val_loss)
        if mean_val_dice > self.best_val_dice:
            self.best_val_dice = mean_val_dice
            self.best_val_epoch = self.current_epoch
        print(f"\ncurrent epoch: {self.current_epoch} mean val dice: {mean_val_dice:.4f}"
              f"\ncurrent mean loss: {mean_val_loss:.4f}"
              f"\nbest mean dice: {self.best_val_dice:.4f} at epoch: {self.best_val_epoch}")
        return {"val_dice": mean_val_dice, "val_loss": mean_val_loss}

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, path: Path, epoch: int = 0, extra: Optional[dict] = None):
        """Lightning-compatible ``.ckpt``: ``state_dict`` keys ``_model.model.<MONAI path>`` and
        ``hyper_parameters`` = constructor kwargs (reference ``:112, :503-509, :564-574``)."""
        sd = {k: v.detach().cpu().clone() for k, v in self.state_dict().items()}
        ckpt = {"state_dict": sd, "hyper_parameters": dict(self.hparams), "epoch": epoch,
                "pytorch-lightning_version": "2.0.0-compatible (segmantic_amd)"}
        if extra:
            ckpt.update(extra)
        torch.save(ckpt, str(path))

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, **overrides) -> "Net":
        ckpt = torch.load(str(checkpoint_path), map_location="cpu", weights_only=False)
        if "state_dict" not in ckpt:
            # scripts/extract_unet.py output: bare inner-UNet state dict (keys "model....")
            sd = {"_model." + k: v for k, v in ckpt.items()}
            hp = {}
        else:
            # a Lightning checkpoint of the reference's Net may carry entries besides the network
            # (loss / metric buffers such as ``loss_function.class_weight``): only ``_model.*`` is ours
            sd = {k: v for k, v in ckpt["state_dict"].items() if k.startswith("_model.")}
            hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        if "num_classes" not in hp:
            last = [k for k in sd if k.endswith("2.1.conv.unit0.conv.bias") and k.count("submodule") == 0]
            hp["num_classes"] = int(sd[last[0]].shape[0])
        if "num_channels" not in hp:
            hp["num_channels"] = int(sd["_model.model.0.conv.unit0.conv.weight"].shape[1])
        import inspect
        known = set(inspect.signature(cls.__init__).parameters) - {"self"}
        net = cls(**{k: v for k, v in hp.items() if k in known})
        net.load_state_dict(sd, strict=True)
        return net

    def freeze(self):
        for p in self.parameters():
            p.requires_grad_(False)


# =============================================================================================
# train / predict
# =============================================================================================
def _ckpt_name(output_dir: Path, epoch: int, val_loss: float, val_dice: float) -> Path:
    # Lightning's "{epoch}-{val_loss:.2f}-{val_dice:.4f}" -> "epoch=E-val_loss=L-val_dice=D.ckpt"
    return Path(output_dir) / f"epoch={epoch}-val_loss={val_loss:.2f}-val_dice={val_dice:.4f}.ckpt"


def train(
    *,
    datalist: Path,
    image_dir: Path = None,
    labels_dir: Path = None,
    output_dir: Path,
    checkpoint_file: Path = None,
    num_classes: int = 0,
    num_channels: int = 1,
    spatial_dims: int = 3,
    spatial_size: Sequence[int] = [],
    preprocessing: dict = {},
    augmentation: dict = {},
    augment_intensity: bool = False,
    augment_spatial: bool = False,
    channels: tuple = (16, 32, 64, 128, 256),
    strides: tuple = (2, 2, 2, 2),
    dropout: float = 0.0,
    act: str = "PRELU",
    num_samples: int = 4,
    optimizer=None,
    lr_scheduling=None,
    max_epochs: int = 600,
    early_stop_patience: int = 50,
    mixed_precision: bool = True,
    cache_rate: float = 1.0,
    gpu_ids: list = [0],
    tissue_list: Path = None,
) -> Net:
    """Same keyword-only signature as the reference's ``train`` (``:400-428``): it *is* the
    YAML/JSON config schema of ``segmantic-unet train-config``."""
    from .dataset import PairedDataSet
    from .trainer import fit

    from . import launch
    n_ranks = len(list(gpu_ids or []))
    if n_ranks > 1 and not launch.under_launcher():
        # several gpu_ids: one process per GPU, started from here as children of this process --
        # what pl.Trainer(devices=len(gpu_ids)) does for the reference (:529-538).  The parent has not
        # touched the GPU yet and never will; it hands the call to the ranks as a train-config file.
        call = dict(datalist=datalist, image_dir=image_dir, labels_dir=labels_dir, output_dir=output_dir,
                    checkpoint_file=checkpoint_file, num_classes=num_classes, num_channels=num_channels,
                    spatial_dims=spatial_dims, spatial_size=spatial_size, preprocessing=preprocessing,
                    augmentation=augmentation, augment_intensity=augment_intensity,
                    augment_spatial=augment_spatial, channels=channels, strides=strides, dropout=dropout,
                    act=act, num_samples=num_samples, optimizer=optimizer, lr_scheduling=lr_scheduling,
                    max_epochs=max_epochs, early_stop_patience=early_stop_patience,
                    mixed_precision=mixed_precision, cache_rate=cache_rate, gpu_ids=gpu_ids,
                    tissue_list=tissue_list)
        return _train_in_ranks(n_ranks, call)

    if optimizer is None:
        optimizer = dict(Net.optimizer)
    if lr_scheduling is None:
        lr_scheduling = dict(Net.lr_scheduling)

    if checkpoint_file and Path(checkpoint_file).exists():
        net = Net.load_from_checkpoint(f"{checkpoint_file}", map_location="cpu")
        net.best_val_dice = 0.0
    else:
        if num_classes > 0 and tissue_list:
            raise ValueError("'num_classes' and 'tissue_list' are redundant. Prefer 'num_classes'.")
        if num_classes <= 0:
            if tissue_list:
                tissue_dict = load_tissue_list(tissue_list)
            else:
                tissue_dict = load_decathlon_tissuelist(Path(datalist))
            num_classes = max(tissue_dict.values()) + 1
            if len(tissue_dict) != num_classes:
                raise ValueError("Expecting contiguous labels in range [0,N-1]")
        if num_classes <= 1:
            raise ValueError("'num_classes' is expected to be > 1")
        net = Net(spatial_dims=spatial_dims, num_channels=num_channels, num_classes=num_classes,
                  spatial_size=spatial_size, channels=channels, strides=strides, dropout=dropout,
                  act=act)
    if image_dir and labels_dir:
        net.dataset = PairedDataSet(image_dir=image_dir, labels_dir=labels_dir)
    elif datalist:
        net.dataset = PairedDataSet.load_from_json(datalist)
    else:
        raise ValueError("Either provide a dataset file, or an image_dir, labels_dir pair.")
    net.config_preprocessing = preprocessing
    net.config_augmentation = augmentation
    net.augment_intensity = augment_intensity
    net.augment_spatial = augment_spatial
    net.num_samples = num_samples
    # MONAI-bundle dictionaries (reference prepare_data, :232-262): resolved with the reference's
    # parser context and mapped onto the on-device pipeline; transforms it cannot express raise here
    from ..utils.bundle import ConfigParser, plan_augmentation, plan_preprocessing
    if preprocessing:
        parser = ConfigParser({"image_key": "image", "label_key": "label", "preprocessing": preprocessing})
        parser.parse(True)
        plan = plan_preprocessing(parser.get_parsed_content("preprocessing"))
        if plan is None:
            print("Using default preprocessing")
        else:
            missing = [k for k in ("orientation", "normalize", "crop_foreground") if not plan[k]]
            if missing:
                raise ValueError(f"'preprocessing': the on-device pipeline always runs {missing}; add the "
                                 "corresponding transforms or leave 'preprocessing' empty")
            net.train_spacing = plan["spacing"]
            net.train_spacing_label_nearest = bool(plan["spacing_label_nearest"])
    if augmentation:
        parser = ConfigParser({"image_key": "image", "label_key": "label", "augmentation": augmentation})
        parser.parse(True)
        plan = plan_augmentation(parser.get_parsed_content("augmentation"))
        if plan is not None:
            if plan["num_classes"] is not None and plan["num_classes"] != net.num_classes:
                raise ValueError(f"'augmentation': RandCropByLabelClassesd is configured for {plan['num_classes']} "
                                 f"classes, the network has {net.num_classes}")
            if plan["flip_axes"] and len(plan["flip_axes"]) != net.spatial_dims:
                raise ValueError(f"'augmentation': RandFlipd entries for axes {plan['flip_axes']} but spatial_dims = "
                                 f"{net.spatial_dims} (one entry per spatial axis)")
            if plan.get("pad_size") is not None and plan["spatial_size"] is not None \
                    and list(plan["pad_size"]) != list(plan["spatial_size"]):
                raise ValueError("'augmentation': SpatialPadd and RandCropByLabelClassesd must share one spatial_size")
            net.num_samples = plan["num_samples"]
            net.flip_prob = plan["flip_prob"]
            net.augment_spatial = net.augment_spatial or plan["augment_spatial"]
            net.augment_intensity = net.augment_intensity or plan["augment_intensity"]
            if plan["spatial_size"] and not spatial_size:
                net.spatial_size = [int(v) for v in plan["spatial_size"]]
    net.optimizer = optimizer
    net.lr_scheduling = lr_scheduling
    net.cache_rate = cache_rate
    net.mixed_precision = bool(mixed_precision)

    # process group first: nothing above has made a device call (the runtime environment was set when
    # the package was imported), and init_distributed() binds the rank to its GPU before the first one
    rank, local_rank, world = env_world()
    if world > 1:
        init_distributed(device_index=rank_device_index(gpu_ids, local_rank))
    if not torch.cuda.is_available():
        raise RuntimeError(
            "segmantic_amd.train needs an MI355X: torch.cuda.is_available() is False and there "
            "is no CPU execution path (the reference's gpu_ids=[] CPU mode is served by the "
            "reference itself)")
    output_dir = Path(output_dir)
    output_dir.mkdir(exist_ok=True, parents=True)
    if world > 1:                         # one process per GPU: every rank trains on rank 0's split
        from .trainer import sync_dataset
        net.dataset = sync_dataset(net.dataset)
    if rank == 0:
        (output_dir / "Dataset.json").write_text(net.dataset.dump_dataset())
    fit(net, output_dir=output_dir, max_epochs=max_epochs, early_stop_patience=early_stop_patience,
        gpu_ids=gpu_ids, ckpt_name=_ckpt_name)
    print(f"train completed, best_metric: {net.best_val_dice:.4f} at epoch {net.best_val_epoch}")
    return net


def _train_in_ranks(n_ranks: int, call: dict) -> Optional[Net]:
    """Run ``train(**call)`` as ``n_ranks`` processes (one per entry of ``gpu_ids``) and return the
    best checkpoint's network THIS launch wrote (on the CPU; None when it wrote none -- checkpoints an
    earlier run left in ``output_dir`` are not candidates; the single-process ``train()`` returns its
    live network, whose weights are the last epoch's: load the best file for those too if needed).
    The call travels as a train-config file: the config schema IS the signature (reference ``:400-428``)."""
    import inspect
    import tempfile
    import time

    from ..utils import config
    from ..utils.cli import cast_from_path
    from . import launch
    sig = inspect.signature(train)
    out = Path(call["output_dir"])
    out.mkdir(exist_ok=True, parents=True)
    plain = {}
    for k, v in call.items():
        v = cast_from_path(v, sig.parameters[k])
        plain[k] = list(v) if isinstance(v, tuple) else v
    with tempfile.NamedTemporaryFile("w", suffix=".json", prefix="ranks_", dir=str(out), delete=False) as f:
        f.write(config.dumps(plain, is_json=True))
        cfg = f.name
    started = time.time() - 2.0             # file-system time stamps may be coarser than the clock
    try:
        rc = launch.spawn_ranks(n_ranks, ["-m", "segmantic_amd.commands.monai_unet_cli", "train-config", "-c", cfg])
    finally:
        os.unlink(cfg)
    if rc != 0:
        raise RuntimeError(f"segmantic_amd.train: the {n_ranks}-rank launch exited with code {rc}")
    best = None
    for p in out.glob("epoch=*-val_dice=*.ckpt"):
        if p.stat().st_mtime < started:     # left behind by an earlier run in the same directory
            continue
        m = re.search(r"val_dice=([0-9.]+?)\.ckpt$", p.name)
        if m and (best is None or float(m.group(1)) > best[0]):
            best = (float(m.group(1)), p)
    return Net.load_from_checkpoint(best[1]) if best else None


def _predict_in_ranks(n_ranks: int, call: dict) -> None:
    """``predict(**call)`` as one process per GPU (volumes dealt round-robin, rank 0 writes the tables)."""
    import pickle
    import tempfile

    from . import launch
    with tempfile.NamedTemporaryFile("wb", suffix=".pkl", prefix="predict_", delete=False) as f:
        pickle.dump(call, f)
        path = f.name
    try:
        rc = launch.spawn_ranks(n_ranks, ["-m", "segmantic_amd.seg.monai_unet", "--predict-call", path])
    finally:
        os.unlink(path)
    if rc != 0:
        raise RuntimeError(f"segmantic_amd.predict: the {n_ranks}-rank launch exited with code {rc}")


def predict(
    model_file: Path,
    test_images: List[Path],
    test_labels: Optional[List[Path]] = None,
    output_dir: Path = None,
    tissue_dict: Dict[str, int] = None,
    channels: tuple = (16, 32, 64, 128, 256),
    strides: tuple = (2, 2, 2, 2),
    dropout: float = 0.0,
    spacing: Sequence[float] = [],
    gpu_ids: List[int] = [],
) -> None:
    """reference ``:551-725``: load checkpoint, sliding-window inference per volume, invert the
    pre-processing on the logits, argmax, save; with labels: per-volume / total Dice tables and the
    ``ConfusionMatrixMetric`` table (sensitivity, specificity, precision, accuracy).  Not
    reproduced: the per-volume ``*_confusion.png`` plots (matplotlib; out of scope).
    The class-Dice tables are headed by the K-1 foreground tissue names (the reference prints all K
    names over the K-1 foreground values, one column off)."""
    from . import launch
    from .pipeline import PredictPipeline

    if len(list(gpu_ids or [])) > 1 and not launch.under_launcher():
        # several gpu_ids: the volumes are independent objects -- one process per GPU, started here
        return _predict_in_ranks(len(gpu_ids), dict(
            model_file=model_file, test_images=test_images, test_labels=test_labels, output_dir=output_dir,
            tissue_dict=tissue_dict, channels=channels, strides=strides, dropout=dropout, spacing=spacing,
            gpu_ids=gpu_ids))
    model_file = Path(model_file)
    settings_json = model_file.with_suffix(".json")
    if settings_json.exists():
        print(f"WARNING: Loading legacy model settings from {settings_json}")
        settings = json.loads(settings_json.read_text())
        net = Net.load_from_checkpoint(f"{model_file}", **settings)
    else:
        # The reference always overrides the checkpoint's hparams with these arguments, which
        # makes its CLI unable to load non-default architectures; here an argument left at its
        # default does not override what the checkpoint recorded.
        over = {}
        if tuple(channels) != (16, 32, 64, 128, 256):
            over["channels"] = channels
        if tuple(strides) != (2, 2, 2, 2):
            over["strides"] = strides
        if dropout != 0.0:
            over["dropout"] = dropout
        net = Net.load_from_checkpoint(f"{model_file}", **over)
    num_classes = net.num_classes
    net.freeze()
    net.eval()
    # one process per GPU under torchrun: the volumes (independent objects) are dealt round-robin
    # to the ranks, no data-path collective; rank 0 collects the per-volume scores at the end
    rank, local_rank, world = env_world()
    if world > 1:
        device = torch.device(f"cuda:{rank_device_index(gpu_ids, local_rank)}")
        init_distributed(device_index=device.index)
        torch.cuda.set_device(device)
    else:
        device = make_device(gpu_ids)
    if device.type != "cuda":
        raise RuntimeError("segmantic_amd.predict needs an MI355X (no CPU execution path)")
    net.to(device)

    use_labels = bool(test_labels) and len(test_images) == len(test_labels)
    pipe = PredictPipeline(device=device, spacing=spacing, with_label=use_labels)
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)
        output_dir = Path(output_dir)
    inferer = SlidingWindowInferer(roi_size=net.spatial_size, sw_batch_size=4, device=device)
    dice_metric = DiceMetric(num_classes, include_background=False)
    confusion_metrics = ["sensitivity", "specificity", "precision", "accuracy"]
    conf_matrix = ConfusionMatrixMetric(num_classes, confusion_metrics)

    tissue_names = [f"{i}" for i in range(num_classes)]
    if tissue_dict:
        for name, idx in tissue_dict.items():
            if 0 <= idx < num_classes:
                tissue_names[idx] = name

    def print_table(header, vals, indent="\t"):
        print(indent + "\t".join(header).expandtabs(30))
        print(indent + "\t".join(f"{x}" for x in vals).expandtabs(30))

    all_mean_dice = []
    class_dice_sum, class_dice_cnt = None, None
    per_volume = []            # (index in test_images, per-class Dice row): what a multi-rank run gathers
    with torch.no_grad():
        for i, img_path in enumerate(test_images):
            if i % world != rank:
                continue
            item = pipe.load(img_path, test_labels[i] if use_labels else None)
            val_pred = inferer(item["image"][None], net)             # [1,K,D,H,W] f32 logits
            label_vol = pipe.invert_and_discretize(val_pred[0], item)
            if output_dir:
                pipe.save(label_vol, item, output_dir)
            if use_labels:
                pred_lab = _argmax_labels(val_pred)
                d = dice_metric(pred_lab, item["label"][None].long())
                conf_matrix(pred_lab, item["label"][None].long())
                dn = d.cpu().numpy()
                per_volume.append((i, d.cpu()))
                print("Mean Dice: ", np.nanmean(dn))
                print("Class Dice:")
                print_table(tissue_names[1:], np.squeeze(dn))
                all_mean_dice.append(float(dice_metric.aggregate().item()))
                dz = np.nan_to_num(dn[0], nan=0.0)
                ok = (~np.isnan(dn[0])).astype(np.float64)
                class_dice_sum = dz if class_dice_sum is None else class_dice_sum + dz
                class_dice_cnt = ok if class_dice_cnt is None else class_dice_cnt + ok
        if world > 1:
            import torch.distributed as dist
            parts = [None] * world
            conf_local = torch.cat(conf_matrix._items).cpu() if conf_matrix._items else None
            dist.all_gather_object(parts, (per_volume, class_dice_sum, class_dice_cnt, conf_local))
            if rank != 0:
                return
            conf_matrix._items = [p[3] for p in parts if p[3] is not None]
            # the score file holds the RUNNING aggregate after every volume (reference :664): rebuild it
            # from all ranks' per-volume rows in the order of `test_images`, as one process would have
            running = DiceMetric(num_classes, include_background=False)
            all_mean_dice = []
            for _i, row in sorted((t for p in parts for t in p[0]), key=lambda t: t[0]):
                running._items.append(row)
                all_mean_dice.append(float(running.aggregate().item()))
            sums = [p[1] for p in parts if p[1] is not None]
            class_dice_sum = sum(sums) if sums else None
            class_dice_cnt = sum(p[2] for p in parts if p[2] is not None) if sums else None
        if output_dir is None:
            print("No output path specified, dice scores won't be saved.")
        else:
            np.savetxt(output_dir / f"mean_dice_{model_file.stem}_generalized_score.txt",
                       all_mean_dice, delimiter=",")
        if use_labels:
            print("*" * 80)
            total = float(dice_metric.aggregate().item()) if world == 1 else float(
                np.nanmean(class_dice_sum / np.maximum(class_dice_cnt, 1)))
            print("Total Mean Dice: ", total)
            print("Total Class Dice:")
            print_table(tissue_names[1:], class_dice_sum / np.maximum(class_dice_cnt, 1))
            print("Total Conf. Matrix Metrics:")
            print_table(confusion_metrics, (float(x) for x in conf_matrix.aggregate()))
    # the sliding-window driver keeps its prediction cache (up to ~23 GB per device) across the volumes of a call;
    # a process that goes on to train must not find it pinned beside its activations
    from .inferers import release_workspaces
    release_workspaces()


def _argmax_labels(logits: torch.Tensor) -> torch.Tensor:
    """[B,K,D,H,W] -> [B,1,D,H,W] int32 via the HIP argmax kernel (first max wins)."""
    lg = as_ndhwc(logits)
    lab = torch.empty(lg.shape[:4], dtype=torch.int32, device=lg.device)
    ops.argmax(lg, lab)
    return lab.unsqueeze(1)


# =============================================================================================
# cross validation / ensembles (reference ``:728-1004``): orchestration over train() / predict()
# =============================================================================================
def cross_validate(
    image_dir: Path,
    labels_dir: Path,
    tissue_list: Path,
    output_dir: Path,
    config_files_dir: Path,
    test_image_dir: Path = None,
    test_labels_dir: Path = None,
    num_splits: int = 7,
    gpu_ids: List[int] = [0],
):
    """reference ``:728-831``: k-fold data lists, one ``train-config`` run per (config file, fold)
    in its own process, optional generalisation test of every checkpoint of the fold.

    Two defects of the reference loop are not reproduced: the training subprocess is started with
    an argument list AND ``shell=True`` (which drops the arguments on POSIX), and the test images
    are globbed with ``".nii.gz"`` (matches nothing) instead of ``"*.nii.gz"``."""
    from .dataset import PairedDataSet

    print("Cross-validating")
    output_dir = Path(output_dir)
    output_dir.mkdir(exist_ok=True, parents=True)
    tissue_dict = load_tissue_list(tissue_list)
    print(tissue_dict)
    with_test = bool(test_image_dir and test_labels_dir)
    test_dicts = (PairedDataSet.create_data_dict(image_dir=test_image_dir, labels_dir=test_labels_dir)
                  if with_test else [])
    fold_lists = PairedDataSet.kfold_crossval(
        num_splits=num_splits,
        data_dicts=PairedDataSet.create_data_dict(image_dir=image_dir, labels_dir=labels_dir),
        output_dir=output_dir / "datafolds", test_data_dicts=test_dicts)

    # 1. lay out every (scenario, fold) run: its directory and its own train-config file
    runs: List[Path] = []
    for scenario in sorted(Path(config_files_dir).iterdir()):
        if scenario.suffix not in (".json", ".yml"):
            raise AssertionError(f"suffix: {scenario}")
        runs += [_write_fold_config(scenario, fold, fold_list, output_dir / scenario.stem)
                 for fold, fold_list in enumerate(fold_lists)]

    # 2. train them one after the other, each in a process of its own; 3. score the fold's checkpoints
    test_images = test_labels = None
    if with_test:
        test_image_dir, test_labels_dir = Path(test_image_dir), Path(test_labels_dir)
        assert test_image_dir.is_dir() and test_labels_dir.is_dir()
        test_images, test_labels = sorted(test_image_dir.glob("*.nii.gz")), sorted(test_labels_dir.glob("*.nii.gz"))
        assert len(test_images) == len(test_labels)
    for fold_config in runs:
        fold_dir = fold_config.parent
        print(fold_dir)
        print("start training")
        print(f"training finished : {_train_config_in_child(fold_config) == 0}")
        for ckpt in sorted(fold_dir.glob("*.ckpt")) if with_test else ():
            print("start prediction")
            predict(model_file=ckpt, output_dir=fold_dir, test_images=test_images, test_labels=test_labels,
                    tissue_dict=tissue_dict, dropout=0.0, spacing=[1, 1, 1], gpu_ids=gpu_ids)


def _write_fold_config(scenario: Path, fold: int, fold_list: Path, scenario_dir: Path) -> Path:
    """the scenario's train-config with its data list / output directory pointed at fold ``fold``; returns the
    file written (``<scenario_dir>/<fold>/config.{json,yml}``, same syntax as the scenario file)"""
    from ..utils import config

    is_json = scenario.suffix.lower() == ".json"
    fold_dir = scenario_dir / str(fold)
    fold_dir.mkdir(exist_ok=True, parents=True)
    args: dict = config.loads(scenario.read_text(), is_json=is_json)
    for key in ("image_dir", "labels_dir"):
        args.pop(key, None)
    args.update(datalist=str(fold_list), output_dir=str(fold_dir))
    out = fold_dir / ("config.json" if is_json else "config.yml")
    out.write_text(config.dumps(args, is_json=is_json))
    return out


def _train_config_in_child(config_file: Path) -> int:
    """``segmantic-unet train-config -c <file>`` as a child process (argument list, no shell); the package may be
    used from a source tree, so the child gets this tree on its PYTHONPATH.  Returns the exit code."""
    import subprocess
    import sys

    env = dict(os.environ)
    here = str(Path(__file__).resolve().parents[2])
    env["PYTHONPATH"] = os.pathsep.join([here] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    cmd = [sys.executable, "-m", "segmantic_amd.commands.monai_unet_cli", "train-config", "-c", str(config_file)]
    return subprocess.run(cmd, cwd=os.fspath(config_file.parent), env=env).returncode


def _one_hot_logits(labels: torch.Tensor, num_classes: int) -> torch.Tensor:
    """[D,H,W] int32 labels -> [K,D,H,W] f32 one-hot (so the label volume can take the same
    inverse chain as logits: trilinear in class-probability space, then argmax)."""
    out = torch.zeros((num_classes,) + tuple(labels.shape), dtype=torch.float32, device=labels.device)
    out.scatter_(0, labels.long().clamp_(0, num_classes - 1)[None], 1.0)
    return out


def ensemble_creator(
    model_files: List[Path],
    test_images: List[Path],
    test_labels: Optional[List[Path]] = None,
    output_dir: Path = None,
    tissue_dict: Dict[str, int] = None,
    spacing: Sequence[float] = [],
    combination_mode="select_best",
    candidate_per_tissue_path: Optional[Path] = None,
    gpu_ids: List[int] = [],
):
    """reference ``:848-1004``: every model predicts each volume with sliding windows of 96^3 at
    overlap 0.5 (``:840-842``); the predictions are combined on the device (``csrc/ensemble.hip``):
    ``mean`` (logits weighted by the ``val_dice`` in the checkpoint file name), ``vote`` (majority
    of the per-model arg-max labels) or ``select_best`` (per tissue, the labels of the model named
    in ``candidate_per_tissue_path``); the result is carried back through the inverse
    pre-processing and saved as ``<stem>.nii.gz``."""
    from ..utils import config
    from .pipeline import PredictPipeline

    mode = getattr(combination_mode, "value", combination_mode)
    if mode not in ("mean", "vote", "select_best"):
        raise ValueError(f"unknown combination mode {combination_mode}")
    if mode == "select_best":
        if candidate_per_tissue_path is None:
            raise ValueError("When using the 'select_best'-mode, candidate_per_tissue_path needs to be specified.")
        if tissue_dict is None:
            raise RuntimeError("'select_best' mode requires a tissue list")
    if not model_files:
        raise ValueError("ensemble_creator needs at least one checkpoint")
    device = make_device(gpu_ids)
    if device.type != "cuda":
        raise RuntimeError("segmantic_amd.ensemble_creator needs an MI355X (no CPU execution path)")
    models = [Net.load_from_checkpoint(str(p)) for p in model_files]
    for m in models:
        m.freeze()
        m.eval()
        m.to(device)
    num_classes = models[0].num_classes
    use_labels = bool(test_labels) and len(test_images) == len(test_labels)
    pipe = PredictPipeline(device=device, spacing=spacing, with_label=use_labels)
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)
        output_dir = Path(output_dir)
    inferer = SlidingWindowInferer(roi_size=(96, 96, 96), sw_batch_size=4, overlap=0.5)
    weights = None
    label_model = None
    if mode == "mean":   # validation metric of each checkpoint as its weight (reference :924-927)
        weights = [float(Path(p).stem.split("-")[-1].split("=")[1]) for p in model_files]
    if mode == "select_best":
        name_model = config.load(config_file=Path(candidate_per_tissue_path))
        label_model = {int(tissue_dict[name]): int(idx) for name, idx in name_model.items()}
    saved = []
    with torch.no_grad():
        for i, img_path in enumerate(test_images):
            item = pipe.load(img_path, test_labels[i] if use_labels else None)
            x = item["image"][None]
            if mode == "mean":
                outs = [inferer(x, m).contiguous().clone() for m in models]      # [1,K,D,H,W] views
                comb = torch.empty_like(outs[0])
                ops.ensemble_mean(outs, weights, comb)
                combined = comb[0]
            else:
                labs = [_argmax_labels(inferer(x, m))[0, 0].contiguous() for m in models]
                out = torch.empty_like(labs[0])
                if mode == "vote":
                    ops.ensemble_vote(labs, out)
                else:
                    ops.ensemble_select(labs, label_model, out)
                combined = _one_hot_logits(out, num_classes)
            label_vol = pipe.invert_and_discretize(combined, item)
            if output_dir:
                saved.append(pipe.save(label_vol, item, output_dir))
    return saved


if __name__ == "__main__":      # rank entry of _predict_in_ranks
    import pickle
    import sys
    if len(sys.argv) == 3 and sys.argv[1] == "--predict-call":
        with open(sys.argv[2], "rb") as _f:
            predict(**pickle.load(_f))
    else:
        raise SystemExit("usage: python -m segmantic_amd.seg.monai_unet --predict-call FILE")
