"""Oracle (test infrastructure): torch.nn CPU restatement of the reference's network, loss, step.

Follows reference ``src/segmantic/seg/monai_unet.py``:
  * ``:114-124``  UNet(spatial_dims, in, out, channels, strides, dropout, num_res_units=2,
                  norm=Norm.BATCH, act=act)           -> ``RefUNet``
  * ``:128``      DiceLoss(to_onehot_y=True, softmax=True)   -> ``ref_dice_loss``
  * ``:339-348``  forward -> zero_grad -> loss -> backward -> optimizer.step  -> ``ref_train_step``
  * ``:299-304``  torch.optim.Adam(lr, amsgrad)

MONAI (unpinned dependency, absent here) semantics restated from its published source:
``monai.networks.nets.UNet._create_block`` recursion, ``ResidualUnit`` (conv units + residual
conv k3/stride-s when strided, k1 when only channels change, identity otherwise),
``Convolution`` = conv -> ADN("NDA": BatchNorm -> Dropout -> PReLU), transposed conv with
``padding=1, output_padding=stride-1``, ``SkipConnection`` = cat([x, sub(x)], dim=1).
State-dict keys are MONAI's (``model.0.conv.unit0.conv.weight`` ...).  PARITY UNPINNED for
values (see package docstring); parameter count and key layout are pinned.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv_nd(dims: int):
    return {2: nn.Conv2d, 3: nn.Conv3d}[dims]


def _convT_nd(dims: int):
    return {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}[dims]


def _bn_nd(dims: int):
    return {2: nn.BatchNorm2d, 3: nn.BatchNorm3d}[dims]


def _act(act: str) -> nn.Module:
    a = act.upper()
    if a == "PRELU":
        return nn.PReLU()  # one scalar alpha, init 0.25
    if a == "RELU":
        return nn.ReLU()
    if a == "LEAKYRELU":
        return nn.LeakyReLU()
    raise ValueError(f"unsupported act {act}")


class _ADN(nn.Sequential):
    """MONAI ADN with ordering "NDA": N (BatchNorm) -> D (Dropout p) -> A (activation)."""

    def __init__(self, dims: int, ch: int, act: str, dropout: float):
        super().__init__()
        self.add_module("N", _bn_nd(dims)(ch))
        if dropout is not None:
            self.add_module("D", nn.Dropout(dropout))
        self.add_module("A", _act(act))


class _Convolution(nn.Sequential):
    """MONAI ``Convolution``: conv (or transposed conv) followed by ADN unless ``conv_only``."""

    def __init__(self, dims, cin, cout, stride, kernel, act, dropout, conv_only=False,
                 transposed=False):
        super().__init__()
        pad = (kernel - 1) // 2
        if transposed:
            conv = _convT_nd(dims)(cin, cout, kernel, stride, pad, output_padding=stride - 1,
                                   bias=True)
        else:
            conv = _conv_nd(dims)(cin, cout, kernel, stride, pad, bias=True)
        self.add_module("conv", conv)
        if not conv_only:
            self.add_module("adn", _ADN(dims, cout, act, dropout))


class _ResidualUnit(nn.Module):
    def __init__(self, dims, cin, cout, stride, kernel, subunits, act, dropout,
                 last_conv_only=False):
        super().__init__()
        self.conv = nn.Sequential()
        self.residual: nn.Module = nn.Identity()
        sc, ss = cin, stride
        subunits = max(1, subunits)
        for su in range(subunits):
            conv_only = last_conv_only and su == subunits - 1
            self.conv.add_module(
                f"unit{su:d}",
                _Convolution(dims, sc, cout, ss, kernel, act, dropout, conv_only=conv_only))
            sc, ss = cout, 1
        if stride != 1 or cin != cout:
            rk, rp = kernel, (kernel - 1) // 2
            if stride == 1:
                rk, rp = 1, 0
            self.residual = _conv_nd(dims)(cin, cout, rk, stride, rp, bias=True)

    def forward(self, x):
        res = self.residual(x)
        cx = self.conv(x)
        return cx + res


class _Skip(nn.Module):
    def __init__(self, submodule):
        super().__init__()
        self.submodule = submodule

    def forward(self, x):
        return torch.cat([x, self.submodule(x)], dim=1)


class RefUNet(nn.Module):
    """torch.nn restatement of ``monai.networks.nets.UNet`` with num_res_units=2, BATCH, act."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int,
                 channels: Sequence[int] = (16, 32, 64, 128, 256),
                 strides: Sequence[int] = (2, 2, 2, 2), dropout: float = 0.0,
                 act: str = "PRELU", num_res_units: int = 2, kernel_size: int = 3):
        super().__init__()
        if len(channels) < 2:
            raise ValueError("the length of `channels` should be no less than 2.")
        if len(strides) < len(channels) - 1:
            raise ValueError("the length of `strides` should equal to `len(channels) - 1`.")
        self.dimensions = spatial_dims
        self.in_channels = in_channels
        self.out_channels = out_channels
        d, k, a, p, nr = spatial_dims, kernel_size, act, dropout, num_res_units

        def down(cin, cout, s):
            return _ResidualUnit(d, cin, cout, s, k, nr, a, p)

        def up(cin, cout, s, is_top):
            conv = _Convolution(d, cin, cout, s, k, a, p, conv_only=False, transposed=True)
            ru = _ResidualUnit(d, cout, cout, 1, k, 1, a, p, last_conv_only=is_top)
            return nn.Sequential(conv, ru)

        def block(inc, outc, chs, sts, is_top):
            c, s = chs[0], sts[0]
            if len(chs) > 2:
                sub = block(c, c, chs[1:], sts[1:], False)
                upc = c * 2
            else:
                sub = down(c, chs[1], 1)  # bottom layer
                upc = c + chs[1]
            return nn.Sequential(down(inc, c, s), _Skip(sub), up(upc, outc, s, is_top))

        self.model = block(in_channels, out_channels, list(channels), list(strides), True)

    def forward(self, x):
        return self.model(x)


# --------------------------------------------------------------------------------------------
# build-owned deterministic filler (no torch RNG so fixtures are reproducible anywhere)
# --------------------------------------------------------------------------------------------
def _hash_uniform(n: int, seed: int) -> np.ndarray:
    """n floats in [-1, 1) from a 64-bit counter hash (splitmix64 finaliser)."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0).astype(np.float32)


def deterministic_fill_(module: nn.Module, seed: int = 0) -> nn.Module:
    """Fill every parameter / buffer of ``module`` from the counter hash.

    conv weights ~ U(-b, b) with b = sqrt(3 / fan_in) (unit-gain variance preserving), biases
    U(-0.1, 0.1), BN gamma in [0.75, 1.25], BN beta U(-0.1, 0.1), running_mean U(-0.1, 0.1),
    running_var in [0.75, 1.25], PReLU alpha in [0.15, 0.35].
    """
    sd = module.state_dict()
    for i, (k, v) in enumerate(sd.items()):
        if not v.dtype.is_floating_point:
            continue
        u = torch.from_numpy(_hash_uniform(v.numel(), seed * 1000 + i + 1)).reshape(v.shape)
        leaf = k.rsplit(".", 1)[-1]
        if ".adn.N." in k:
            if leaf in ("weight", "running_var"):
                v.copy_(1.0 + 0.25 * u)
            else:
                v.copy_(0.1 * u)
        elif ".adn.A." in k:
            v.copy_(0.25 + 0.1 * u)
        elif leaf == "bias":
            v.copy_(0.1 * u)
        else:  # conv / convT / residual weight
            if v.dim() >= 3:
                rf = int(np.prod(v.shape[2:]))
                # Conv: [Cout, Cin, k..] fan_in = Cin*rf; ConvT: [Cin, Cout, k..], each output
                # voxel sees ~rf/2^d taps -> use Cin*rf/8 for 3-D stride-2 transposed convs.
                is_t = ".2.0.conv." in k or k.endswith("2.0.conv.weight")
                fan_in = v.shape[0] * rf / (2 ** (v.dim() - 2)) if is_t else v.shape[1] * rf
                v.copy_(float(np.sqrt(3.0 / fan_in)) * u)
            else:
                v.copy_(u)
    module.load_state_dict(sd)
    return module


# --------------------------------------------------------------------------------------------
# loss / step
# --------------------------------------------------------------------------------------------
def ref_dice_loss(logits: torch.Tensor, labels: torch.Tensor, smooth_nr: float = 1e-5,
                  smooth_dr: float = 1e-5) -> torch.Tensor:
    """MONAI DiceLoss(to_onehot_y=True, softmax=True) with defaults (include_background=True,
    squared_pred=False, jaccard=False, batch=False, reduction="mean").

    logits [B,K,*sp] float, labels [B,1,*sp] (integer-valued, any float/int dtype).
    """
    k = logits.shape[1]
    p = torch.softmax(logits, 1)
    t = F.one_hot(labels[:, 0].long(), k).movedim(-1, 1).to(p.dtype)
    axes = list(range(2, logits.dim()))
    inter = (p * t).sum(axes)
    den = t.sum(axes) + p.sum(axes)
    f = 1.0 - (2.0 * inter + smooth_nr) / (den + smooth_dr)
    return f.mean()


def ref_train_step(net: nn.Module, opt: torch.optim.Optimizer, images: torch.Tensor,
                   labels: torch.Tensor):
    """reference ``training_step`` order (monai_unet.py:339-348). Returns (logits, loss)."""
    out = net(images)
    opt.zero_grad()
    loss = ref_dice_loss(out, labels)
    loss.backward()
    opt.step()
    return out.detach(), loss.detach()


def synthetic_batch(batch: int, size: int, num_classes: int, seed: int = 0, dims: int = 3):
    """Synthetic image / blob-like label pair used by tests, smoke and bench.

    Images: hash-uniform noise rescaled to ~unit variance.  Labels: nested shells around a
    per-sample centre so that every class is present and Dice is meaningful.
    """
    shape = (batch, 1) + (size,) * dims
    n = int(np.prod(shape))
    img = _hash_uniform(n, 1234 + seed).reshape(shape) * np.float32(np.sqrt(3.0))
    ax = [np.arange(size, dtype=np.float32)] * dims
    grid = np.meshgrid(*ax, indexing="ij")
    lab = np.zeros(shape, np.float32)
    for b in range(batch):
        c = [size * (0.35 + 0.3 * ((b * 7 + i * 3 + seed) % 5) / 4.0) for i in range(dims)]
        r = np.sqrt(sum((g - ci) ** 2 for g, ci in zip(grid, c)))
        lab[b, 0] = np.clip(np.floor(num_classes * (1.0 - r / (0.75 * size))), 0,
                            num_classes - 1)
    return torch.from_numpy(img), torch.from_numpy(lab)


def state_dict_spec(sd) -> "OrderedDict[str, tuple]":
    return OrderedDict((k, tuple(v.shape)) for k, v in sd.items())
