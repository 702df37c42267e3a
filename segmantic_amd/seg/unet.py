"""3D residual UNet on the HIP kernels of libsegmi: parameter container + execution engine.

Mirrors the network the reference builds at ``src/segmantic/seg/monai_unet.py:114-124``
(``monai.networks.nets.UNet(..., num_res_units=2, norm=Norm.BATCH, act=act)``):

* ``UNetParams`` is a pure parameter container whose ``state_dict()`` keys/shapes are MONAI's
  (``model.0.conv.unit0.conv.weight`` ...; 148 tensors for the default 5-level net) so that
  reference checkpoints load unchanged.
* ``UNetEngine`` owns the MI355X execution plan: NDHWC activations, concat-by-offset skip
  buffers, fused conv epilogues (bias + BN statistics in training; folded BN + PReLU +
  residual in eval), explicit hand-scheduled backward (dgrad via the conv / transposed-conv
  kernels with transformed weight packs, MFMA wgrad, two-pass BN/PReLU backward), one flat f32
  parameter arena and one flat gradient arena (fused optimiser, bucketed all-reduce).

No MONAI, no torch.nn compute ops, no autograd graph inside the network: every FLOP runs in a
hand-written gfx950 kernel reached through the C-ABI.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import ops
from . import streams


# =============================================================================================
# parameter container (MONAI key layout)
# =============================================================================================
class _Box(nn.Module):
    """Name-space node: holds children / parameters, computes nothing."""


def _conv_params(cin: int, cout: int, k: int, dims: int, transposed: bool) -> _Box:
    m = _Box()
    shape = ((cin, cout) if transposed else (cout, cin)) + (k,) * dims
    w = torch.empty(shape)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))          # torch's default conv init
    fan_in = shape[1] * k ** dims                         # torch: weight.size(1) * receptive
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
    b = torch.empty(cout).uniform_(-bound, bound)
    m.weight = nn.Parameter(w)
    m.bias = nn.Parameter(b)
    return m


# MONAI activation names this build maps onto its PReLU-shaped kernels: y = x > 0 ? x : slope * x
# with a learnable scalar slope (PRELU, the reference default) or a fixed one.
FIXED_SLOPE = {"RELU": 0.0, "LEAKYRELU": 0.01}


def _adn_params(ch: int, act: str = "PRELU") -> _Box:
    adn = _Box()
    n = _Box()
    n.weight = nn.Parameter(torch.ones(ch))
    n.bias = nn.Parameter(torch.zeros(ch))
    n.register_buffer("running_mean", torch.zeros(ch))
    n.register_buffer("running_var", torch.ones(ch))
    n.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
    adn.add_module("N", n)
    if act == "PRELU":                       # torch.nn.PReLU(): one learnable slope, init 0.25
        a = _Box()
        a.weight = nn.Parameter(torch.full((1,), 0.25))
        adn.add_module("A", a)               # ReLU / LeakyReLU have no parameters (as in MONAI's ADN)
    return adn


def _convolution_params(cin, cout, k, dims, conv_only=False, transposed=False, act="PRELU") -> _Box:
    m = _Box()
    m.add_module("conv", _conv_params(cin, cout, k, dims, transposed))
    if not conv_only:
        m.add_module("adn", _adn_params(cout, act))
    return m


def _residual_unit_params(cin, cout, stride, k, dims, subunits, last_conv_only=False, act="PRELU") -> _Box:
    m = _Box()
    conv = _Box()
    sc = cin
    for su in range(max(1, subunits)):
        conv_only = last_conv_only and su == max(1, subunits) - 1
        conv.add_module(f"unit{su:d}", _convolution_params(sc, cout, k, dims, conv_only, act=act))
        sc = cout
    m.add_module("conv", conv)
    if stride != 1 or cin != cout:
        rk = k if stride != 1 else 1
        m.add_module("residual", _conv_params(cin, cout, rk, dims, False))
    return m


class UNetParams(nn.Module):
    """Parameter container with MONAI ``UNet`` attribute names (``.model`` Sequential tree)."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int,
                 channels: Sequence[int], strides: Sequence[int], num_res_units: int = 2,
                 kernel_size: int = 3, act: str = "PRELU", dropout: float = 0.0):
        super().__init__()
        if len(channels) < 2:
            raise ValueError("the length of `channels` should be no less than 2.")
        if len(strides) < len(channels) - 1:
            raise ValueError("the length of `strides` should equal to `len(channels) - 1`.")
        if num_res_units < 1:
            raise ValueError("segmantic builds its UNet with num_res_units=2")
        act = str(act).upper()
        if act != "PRELU" and act not in FIXED_SLOPE:
            raise NotImplementedError(f"segmantic_amd implements act in PRELU / RELU / LEAKYRELU, not {act}")
        self.act = act
        self.dropout = float(dropout or 0.0)
        if not 0.0 <= self.dropout < 1.0:
            raise ValueError("dropout must be in [0, 1)")
        self.dimensions = spatial_dims
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.channels = tuple(int(c) for c in channels)
        self.strides = tuple(int(s) for s in strides)
        d, k, nr = spatial_dims, kernel_size, num_res_units

        def block(inc, outc, chs, sts, is_top):
            c, s = chs[0], sts[0]
            seq = _Box()
            if len(chs) > 2:
                sub = block(c, c, chs[1:], sts[1:], False)
                upc = c * 2
            else:
                sub = _residual_unit_params(c, chs[1], 1, k, d, nr, act=act)
                upc = c + chs[1]
            skip = _Box()
            skip.add_module("submodule", sub)
            up = _Box()
            up.add_module("0", _convolution_params(upc, outc, k, d, transposed=True, act=act))
            up.add_module("1", _residual_unit_params(outc, outc, 1, k, d, 1, last_conv_only=is_top, act=act))
            seq.add_module("0", _residual_unit_params(inc, c, s, k, d, nr, act=act))
            seq.add_module("1", skip)
            seq.add_module("2", up)
            return seq

        self.add_module("model", block(in_channels, out_channels, list(self.channels),
                                       list(self.strides), True))


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# =============================================================================================
# execution engine
# =============================================================================================
class _Conv:
    """One (transposed) convolution: parameter views + cached weight packs."""

    def __init__(self, eng: "UNetEngine", prefix: str, cin, cout, k, stride, transposed=False):
        self.eng, self.prefix = eng, prefix
        self.cin, self.cout, self.k, self.stride, self.transposed = cin, cout, k, stride, transposed
        self.w = eng.param(prefix + ".weight")
        self.b = eng.param(prefix + ".bias")
        self.gw = eng.grad(prefix + ".weight")
        self.gb = eng.grad(prefix + ".bias")
        self._packs: Dict[tuple, Tuple[int, Optional[torch.Tensor]]] = {}
        self._fold: Dict[str, tuple] = {}
        # training pair (segmi_conv3d_fwd_split_act): the residual convolution of this first subunit's unit, when the
        # two can run as ONE launch over a pack with both weight sets (set by UNetEngine._make_ru)
        self.pair_with: Optional["_Conv"] = None
        eng._convs.append(self)

    @property
    def mfma(self) -> bool:
        return ops.mfma_ok(self.cin, self.cout)

    def _pack(self, tag, kind, cin_k, cout_k, k, scale=None):
        if not self.mfma:
            return None
        key = (tag, self.eng.dtype)
        ver = self.eng.weights_version
        if self.eng._packed_version != ver:
            self.eng._repack_all()
        self.eng._await_packs()
        if self in self.eng._carry_convs:
            self.eng.sync_weights()          # its update + re-pack of the last step ran on the side stream
        hit = self._packs.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        buf = hit[1] if hit is not None else None
        buf = ops.wpack(self.eng.dtype, kind, self.w, cin_k, cout_k, k, scale=scale, out=buf)
        self._packs[key] = (ver, buf)
        return buf

    # forward operand: (kind, cin_k, cout_k, ksize)
    def fwd_geom(self):
        if self.transposed:
            return (2, self.cin, self.cout, 3)
        return (0, self.cin, self.cout, self.k)

    # input-gradient operand (the kernel that consumes it sees cin_k = cout, cout_k = cin)
    def dgrad_geom(self):
        if self.transposed:   # dgrad of convT = stride-2 conv over dy with the weight as is
            return (0, self.cout, self.cin, 3)
        if self.stride == 2:  # dgrad of a stride-2 conv = transposed-conv kernel, weight as is
            return (2, self.cout, self.cin, 3)
        return (1, self.cout, self.cin, self.k)

    def fwd_pack(self):
        return self._pack("fwd", *self.fwd_geom())

    def dgrad_pack(self):
        return self._pack("dgrad", *self.dgrad_geom())

    def pair_pack(self):
        """[this conv's weight | pair_with's weight] along the output channels (re-packed with the others after every
        optimiser step: ``UNetEngine._repack_all``, two sources per descriptor)"""
        key = ("pair", self.eng.dtype)
        ver = self.eng.weights_version
        if self.eng._packed_version != ver:
            self.eng._repack_all()
        self.eng._await_packs()
        hit = self._packs.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        cat = torch.cat([self.w, self.pair_with.w], 0).contiguous()
        buf = ops.wpack(self.eng.dtype, 0, cat, self.cin, 2 * self.cout, self.k, out=hit[1] if hit is not None else None)
        self._packs[key] = (ver, buf)
        return buf

    def pair_dgrad_pack(self):
        """input-gradient operand of the pair: ONE transposed convolution over [dy of this conv | dy of pair_with]
        (2 cout input channels)"""
        key = ("pair_dgrad", self.eng.dtype)
        ver = self.eng.weights_version
        if self.eng._packed_version != ver:
            self.eng._repack_all()
        self.eng._await_packs()
        hit = self._packs.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        cat = torch.cat([self.w, self.pair_with.w], 0).contiguous()
        buf = ops.wpack(self.eng.dtype, 2, cat, 2 * self.cout, self.cin, 3, out=hit[1] if hit is not None else None)
        self._packs[key] = (ver, buf)
        return buf

    # eval mode: BatchNorm folded into the weights (scale) and bias
    def folded(self, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor]):
        ver = self.eng.weights_version
        hit = self._fold.get("f")
        if hit is not None and hit[0] == ver:
            return hit[1], hit[2], hit[3]
        if scale is None:
            bias, wsrc, pack = self.b, self.w, self.fwd_pack()
        else:
            bias = torch.addcmul(shift, self.b, scale)  # b*scale + shift  (C-element prep)
            if self.mfma:
                kind = 2 if self.transposed else 0
                pack = ops.wpack(self.eng.dtype, kind, self.w, self.cin, self.cout,
                                 3 if self.transposed else self.k, scale=scale)
                wsrc = None
            else:
                pack = None
                shp = [1] * self.w.dim()
                shp[1 if self.transposed else 0] = self.cout
                wsrc = (self.w * scale.view(shp)).contiguous()
        self._fold["f"] = (ver, pack, wsrc, bias)
        return pack, wsrc, bias


class _BN:
    def __init__(self, eng: "UNetEngine", prefix: str, c: int):
        self.eng, self.prefix, self.c = eng, prefix, c
        self.gamma = eng.param(prefix + ".N.weight")
        self.beta = eng.param(prefix + ".N.bias")
        if eng.net.act == "PRELU":
            self.alpha = eng.param(prefix + ".A.weight")
        else:                                  # fixed slope: a device scalar the kernels read
            self.alpha = eng.fixed_slope
        self.g_gamma = eng.grad(prefix + ".N.weight")
        self.g_beta = eng.grad(prefix + ".N.bias")
        self.g_alpha = eng.grad(prefix + ".A.weight") if eng.net.act == "PRELU" else None
        self.rm = eng.buffer(prefix + ".N.running_mean")
        self.rv = eng.buffer(prefix + ".N.running_var")
        self.nbt = eng.buffer(prefix + ".N.num_batches_tracked")
        dev = eng.device
        self.mean = torch.empty(c, device=dev)
        self.invstd = torch.empty(c, device=dev)
        self.scale = torch.empty(c, device=dev)
        self.shift = torch.empty(c, device=dev)
        self.coef = torch.empty((2, c), device=dev)
        self._eval: Optional[tuple] = None
        self.index = len(eng._bns)
        eng._bns.append(self)

    def drop(self) -> tuple:
        """(p, seed) of this layer's ADN dropout for the current training step: forward and the two
        backward passes of a step see the same seed (the mask is recomputed, never stored)."""
        e = self.eng
        if e.dropout_p <= 0.0:
            return (0.0, 0)
        return (e.dropout_p, (e.dropout_seed * 0x9E3779B1 + e._drop_step * 8191 + self.index * 131071) & 0xFFFFFFFF)

    def eval_affine(self):
        ver = self.eng.weights_version
        if self._eval is None or self._eval[0] != ver:
            sc = torch.empty(self.c, device=self.eng.device)
            sh = torch.empty(self.c, device=self.eng.device)
            ops.bn_eval_affine(self.gamma, self.beta, self.rm, self.rv, self.eng.eps, sc, sh)
            self._eval = (ver, sc, sh)
        return self._eval[1], self._eval[2]


class UNetEngine:
    """Execution plan for one ``UNetParams`` on one MI355X."""

    eps = 1e-5
    momentum = 0.1

    def __init__(self, params: UNetParams, device: torch.device, dtype: torch.dtype):
        if params.dimensions not in (2, 3):
            raise NotImplementedError("segmantic_amd: spatial_dims must be 2 or 3")
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("compute dtype must be float32 or bfloat16")
        if any(s not in (1, 2) for s in params.strides):
            raise NotImplementedError("segmantic_amd: strides must be 1 or 2")
        self.net = params
        self.device = torch.device(device)
        self.dtype = dtype
        self.weights_version = 0
        self._packed_version = -1
        self._convs: list = []
        self._bns: list = []
        self.dropout_p = float(getattr(params, "dropout", 0.0) or 0.0)
        self.dropout_seed = 0
        self._drop_step = 0
        self._wbatch = None
        self.training = True
        self._bufs: Dict[str, torch.Tensor] = {}
        self._scratch: Dict[str, torch.Tensor] = {}
        self._lane = 0
        self._saved: Dict[str, torch.Tensor] = {}
        self.timings: Dict[str, list] = {}
        self._carry_convs: set = set()
        self.fixed_slope = torch.full((1,), FIXED_SLOPE.get(params.act, 0.0), dtype=torch.float32,
                                      device=self.device)
        self._build_arena()
        self._build_plan()

    # ------------------------------------------------------------------ arenas
    # Channel padding of the class axis.  The MFMA kernels want channel counts that are multiples
    # of 16; num_classes rarely is (2..10 tissues are typical).  Instead of falling back to the
    # generic direct kernels for the three full-resolution layers that carry K channels (4x slower
    # measured at K = 3), the engine stores those layers with kpad = ceil16(K) channels: extra
    # weights / biases / BN parameters are zero and stay zero (their gradients are exactly 0), extra
    # activations are exactly 0, and softmax / Dice / argmax / blend read only the first K channels
    # of the kpad-strided tensors.  nn.Parameters and checkpoints keep MONAI's shapes: they are
    # (strided) views of the padded arena slots.
    _PADDED = {"model.2.0.conv.weight": (1,), "model.2.0.conv.bias": (0,),
               "model.2.0.adn.N.weight": (0,), "model.2.0.adn.N.bias": (0,),
               "model.2.1.conv.unit0.conv.weight": (0, 1), "model.2.1.conv.unit0.conv.bias": (0,)}
    _PADDED_BUFFERS = ("model.2.0.adn.N.running_mean", "model.2.0.adn.N.running_var")

    def _arena_layout(self, name: str, p) -> Tuple[tuple, tuple]:
        """(shape of the arena slot, index that selects the nn.Parameter's view of it).

        spatial_dims=2: a [Co, Ci, k, k] kernel lives in the centre plane of a [Co, Ci, k, k, k] one
        (zeros elsewhere) so the 3-D kernels compute the 2-D convolution on a depth-1 volume: the
        off-centre taps only ever meet the zero padding, their weights and gradients stay exactly 0."""
        shape = list(p.shape)
        index = [slice(None)] * len(shape)
        for d in self._PADDED.get(name, ()) if self.kpad != self.net.out_channels else ():
            index[d] = slice(0, shape[d])
            shape[d] = self.kpad
        if self.net.dimensions == 2 and p.dim() == 4:
            shape = shape[:2] + [shape[2]] + shape[2:]
            index = index[:2] + [shape[2] // 2] + index[2:]
        return tuple(shape), tuple(index)

    def _build_arena(self):
        k = self.net.out_channels
        self.kpad = k if k % 16 == 0 else (k + 15) // 16 * 16
        named = list(self.net.named_parameters())
        layout = {name: self._arena_layout(name, p) for name, p in named}
        total = sum(int(math.prod(layout[name][0])) for name, _ in named)
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=self.device)
        self._pviews: Dict[str, torch.Tensor] = {}
        self._gviews: Dict[str, torch.Tensor] = {}
        self._galias: Dict[str, torch.Tensor] = {}
        self.param_offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        self._palias: Dict[str, torch.Tensor] = {}
        for name, p in named:
            shape, index = layout[name]
            n = int(math.prod(shape))
            view = self.flat[off:off + n].view(shape)
            g = self.flat_grad[off:off + n].view(shape)
            alias, galias = view[index], g[index]
            alias.copy_(p.data.to(self.device, torch.float32))
            p.data = alias                     # nn.Parameter now aliases the arena
            p.grad = galias
            self._pviews[name] = view
            self._gviews[name] = g
            self._palias[name] = alias
            self._galias[name] = galias
            self.param_offsets[name] = (off, n)
            off += n
        self._bviews: Dict[str, torch.Tensor] = {}
        self._balias: Dict[str, torch.Tensor] = {}
        # the int64 ``num_batches_tracked`` counters share one arena: one add per training step
        nbt = [n for n, _ in self.net.named_buffers() if n.endswith("num_batches_tracked")]
        self._nbt_flat = torch.zeros(max(len(nbt), 1), dtype=torch.int64, device=self.device)
        for name, b in self.net.named_buffers():
            if name in nbt:
                view = self._nbt_flat[nbt.index(name)]
                view.copy_(b.data.to(self.device))
                b.data = view
                full = view
            elif name in self._PADDED_BUFFERS and self.kpad != k:
                full = torch.zeros(self.kpad, dtype=b.dtype, device=self.device)
                if name.endswith("running_var"):
                    full.fill_(1.0)
                full[:k].copy_(b.data.to(self.device))
                b.data = full[:k]
            else:
                b.data = b.data.to(self.device)
                full = b.data
            self._bviews[name] = full
            self._balias[name] = b.data
        self.num_params = total

    def rebind(self):
        """Re-alias parameters after an external ``.to()`` / ``load_state_dict`` replaced data."""
        for name, p in self.net.named_parameters():
            alias = self._palias[name]
            if p.data.data_ptr() != alias.data_ptr():
                alias.copy_(p.data.to(self.device, torch.float32))
                p.data = alias
            p.grad = self._galias[name]
        for name, b in self.net.named_buffers():
            tgt = self._balias[name]
            if b.data.data_ptr() != tgt.data_ptr():
                tgt.copy_(b.data.to(self.device))
                b.data = tgt
        self.weights_version += 1

    def _repack_all(self):
        """One ``segmi_wpack_batch`` launch refreshes the forward and input-gradient operand of
        every MFMA convolution for the current ``weights_version``."""
        ver = self.weights_version
        if self._wbatch is None:
            groups = {False: ([], []), True: ([], [])}         # [carried?] -> (entries, slots)
            for conv in self._convs:
                if not conv.mfma:
                    continue
                entries, slots = groups[conv in self._carry_convs]
                for tag, geom in (("fwd", conv.fwd_geom()), ("dgrad", conv.dgrad_geom())):
                    kind, cin_k, cout_k, k = geom
                    entries.append((kind, conv.w, None, cin_k, cout_k, k))
                    slots.append((conv, (tag, self.dtype)))
                if conv.pair_with is not None and self.pair_train and conv not in self._carry_convs:
                    entries.append((0, conv.w, None, conv.cin, 2 * conv.cout, conv.k, conv.pair_with.w, conv.cout))
                    slots.append((conv, ("pair", self.dtype)))
                    if self.pair_bwd:
                        entries.append((2, conv.w, None, 2 * conv.cout, conv.cin, 3, conv.pair_with.w, conv.cout))
                        slots.append((conv, ("pair_dgrad", self.dtype)))
            self._wbatch = {c: ((ops.WpackBatch(self.dtype, e), sl) if e else (None, [])) for c, (e, sl) in groups.items()}
        for carried in (False, True):
            if carried and self._tail_packed_version == ver:
                continue                   # done by the tail of the last training step (repack_tail)
            if carried:
                self.sync_weights()        # their optimiser update may have run on the weight-gradient stream
            batch, slots = self._wbatch[carried]
            if batch is not None:
                batch.run()
                for (conv, key), buf in zip(slots, batch.packed):
                    conv._packs[key] = (ver, buf)
        self._packed_version = ver

    _tail_packed_version = -1

    def repack_tail(self, ver: int):
        """packs of the carried convolutions for weights version ``ver`` (current stream = the
        weight-gradient stream, right after their optimiser update)"""
        if self._wbatch is None:
            self._packed_version = -1          # first step: the table does not exist yet -> full re-pack later
            return
        batch, slots = self._wbatch[True]
        if batch is not None:
            batch.run()
            for (conv, key), buf in zip(slots, batch.packed):
                conv._packs[key] = (ver, buf)
        self._tail_packed_version = ver

    _pack_ev = None

    def _repack_async(self):
        """Training forward: the one re-pack launch of the step runs on the side stream while the
        main stream does the first layer (Cin = 1: no packed operand); the first MFMA conv waits."""
        if self._packed_version == self.weights_version or not self.overlap_wgrad:
            return
        # (with carried weight gradients the weight-gradient stream is still busy with the last step's
        # tail: the re-pack of all other layers takes a stream of its own)
        main, side = torch.cuda.current_stream(), (self._pack_stream() if self._tail_ev is not None else self._side_stream())
        ev = torch.cuda.Event()
        ev.record(main)                 # after the optimiser step that changed the weights
        side.wait_event(ev)
        with torch.cuda.stream(side):
            self._repack_all()
            self._pack_ev = torch.cuda.Event()
            self._pack_ev.record(side)

    def _await_packs(self):
        if self._pack_ev is not None:
            torch.cuda.current_stream().wait_event(self._pack_ev)
            self._pack_ev = None

    def param(self, key):
        return self._pviews["model." + key]

    def grad(self, key):
        return self._gviews["model." + key]

    def buffer(self, key):
        return self._bviews["model." + key]

    # ------------------------------------------------------------------ plan
    def _build_plan(self):
        chs, sts = list(self.net.channels), list(self.net.strides)
        self.levels = self._make_level("", self.net.in_channels, self.kpad, chs, sts, True)
        # carried weight gradients (see carry_top_wgrad): the up path (transposed conv + unit) of the upper
        # `carry_levels` levels; their parameters (with the BatchNorms between them) are the contiguous END of
        # the arena, in the order a forward reaches them (MONAI's nesting = depth-first order)
        lv, carried = self.levels, []
        while lv is not None and len(carried) < max(1, self.carry_levels):
            carried.append(lv)
            lv = lv.get("sub")
        self._carry_lvls = [id(l) for l in carried]
        self._carry_convs = set()
        for l in carried:
            self._carry_convs |= {l["upconv"]} | {c for c, _ in l["upru"]["units"]}
        pre = tuple(f"model.{l['prefix']}2." for l in carried)
        self.carry_lo = self.param_offsets[f"model.{carried[-1]['prefix']}2.0.conv.weight"][0]
        assert all(off >= self.carry_lo for name, (off, _n) in self.param_offsets.items() if name.startswith(pre)) \
            and all(off < self.carry_lo for name, (off, _n) in self.param_offsets.items() if not name.startswith(pre))

    def _make_ru(self, prefix, cin, cout, stride, subunits, last_conv_only=False):
        units = []
        sc, ss = cin, stride
        for su in range(subunits):
            conv_only = last_conv_only and su == subunits - 1
            conv = _Conv(self, f"{prefix}.conv.unit{su}.conv", sc, cout, 3, ss)
            bn = None if conv_only else _BN(self, f"{prefix}.conv.unit{su}.adn", cout)
            units.append((conv, bn))
            sc, ss = cout, 1
        res = None
        if stride != 1 or cin != cout:
            res = _Conv(self, f"{prefix}.residual", cin, cout, 3 if stride != 1 else 1, stride)
            c0, bn0 = units[0]
            if (len(units) >= 2 and bn0 is not None and c0.mfma and res.mfma and c0.k == 3 and res.k == 3
                    and c0.stride == res.stride == 2):
                c0.pair_with = res
        return {"prefix": prefix, "units": units, "res": res, "cin": cin, "cout": cout,
                "stride": stride}

    def _make_level(self, prefix, inc, outc, chs, sts, is_top):
        p = prefix  # e.g. "" , "1.submodule." ...
        c, s = chs[0], sts[0]
        lvl = {"prefix": p, "c": c, "stride": s, "is_top": is_top, "inc": inc, "outc": outc}
        lvl["down"] = self._make_ru(p + "0", inc, c, s, 2)
        if len(chs) > 2:
            lvl["sub"] = self._make_level(p + "1.submodule.", c, c, chs[1:], sts[1:], False)
            lvl["bottom"] = None
            upc = 2 * c
            lvl["subc"] = c
        else:
            lvl["sub"] = None
            lvl["bottom"] = self._make_ru(p + "1.submodule", c, chs[1], 1, 2)
            upc = c + chs[1]
            lvl["subc"] = chs[1]
        lvl["upc"] = upc
        lvl["upconv"] = _Conv(self, p + "2.0.conv", upc, outc, 3, s, transposed=True)
        lvl["upbn"] = _BN(self, p + "2.0.adn", outc)
        lvl["upru"] = self._make_ru(p + "2.1", outc, outc, 1, 1, last_conv_only=is_top)
        if s != 2:
            raise NotImplementedError("segmantic_amd: the up path implements stride-2 levels")
        return lvl

    # ------------------------------------------------------------------ buffers
    def _buf(self, name, shape, dtype=None) -> torch.Tensor:
        dtype = dtype or self.dtype
        if self._lane:                      # a second in-flight forward owns its own activations
            name = f"lane{self._lane}:{name}"
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(tuple(shape), dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t

    def _scratch_buf(self, name, nbytes) -> torch.Tensor:
        if self._lane:
            name = f"lane{self._lane}:{name}"
        t = self._scratch.get(name)
        if t is None or t.numel() < nbytes:
            t = torch.empty((int(nbytes * 1.25) // 256 + 2) * 256, dtype=torch.uint8,
                            device=self.device)
            self._scratch[name] = t
        return t

    def _fstat(self, rows, c) -> torch.Tensor:
        return self._scratch_buf("stats", rows * 3 * c * 4).view(torch.float32)

    @staticmethod
    def _down_shape(shape, stride):
        n, d, h, w, _ = shape
        f = lambda v: (v + 2 - 3) // stride + 1
        return (n, f(d), f(h), f(w))

    # ------------------------------------------------------------------ primitive steps
    def _pair_ok(self, ru, x, out, dst_name, oshape) -> bool:
        """subunit 0 + residual convolution of a small-Cin unit can share one launch
        (segmi_conv3d_fwd_pair): two or more subunits, both k3 with the same stride"""
        units, rc = ru["units"], ru["res"]
        if len(units) < 2 or units[0][1] is None or rc is None:
            return False
        c0 = units[0][0]
        if c0.mfma or rc.mfma or c0.k != 3 or rc.k != 3 or c0.stride != rc.stride or c0.cout != rc.cout:
            return False
        return ops.conv3d_pair_ok(x, self._buf(dst_name, oshape + (c0.cout,)), out)

    def _tf_ok(self, x, y, conv: _Conv) -> bool:
        """the BatchNorm-apply + PReLU that would produce ``conv``'s input can be done by the conv
        itself while it stages x (segmi_in_affine): no normalised copy of x is ever written"""
        return (self.fuse_bn_apply and self.dtype == torch.bfloat16 and self.dropout_p <= 0.0
                and not conv.transposed and conv.k == 3 and conv.stride == 1 and conv.mfma
                and ops.conv3d_in_affine_ok(x, y, 3, 1))

    merge_eval_pairs = os.environ.get("SEGMI_MERGE_PAIRS", "1") != "0"
    eval_top_fused = False      # set once an eval forward took the fused decoder-top launch (bench.py labels its roofline)
    # inference: transposed conv + conv-only unit of the full-resolution decoder as one launch
    # (csrc/dectop.hip): the 16-channel full-resolution tensor between them never reaches HBM (218 -> 84 MB
    # per 128^3 patch).  Bit-identical to the two launches.  Round 2's version lost (69 vs 59 us per patch);
    # with the producers re-tiled as 16 base voxels x 8 parity classes (round 3) it takes 46-49 us against
    # 58 and the 512^3 benchmark gains 8-9 % (22.9 vs 21.0 volumes/s, same process order, one box).
    # SEGMI_FUSE_EVAL_TOP=0 restores the two launches.
    fuse_eval_top = os.environ.get("SEGMI_FUSE_EVAL_TOP", "1") != "0"

    # Training: the stride-2 first subunit and the residual convolution of an encoder unit read the same input; as
    # two launches of the tile-at-a-time kernel they cost 54 + 54 us (16 -> 32 at 64^3, batch 8), 44 + 44 and 14 + 24
    # one and two levels down.  The eval pairing with BatchNorm statistics on the first half (round 4) runs them as
    # one launch each.  SEGMI_PAIR_TRAIN=0: two launches (A/B).
    pair_train = os.environ.get("SEGMI_PAIR_TRAIN", "1") != "0"
    # ... and their input gradients as ONE transposed convolution over the two output gradients, which then live in
    # the halves of one [.., 2c] buffer: the BatchNorm backward of subunit 0 writes the first half, the producer of
    # the unit's output gradient (the level below) the second.  SEGMI_PAIR_BWD=0: two launches.
    pair_bwd = pair_train and os.environ.get("SEGMI_PAIR_BWD", "1") != "0"

    def _pair_grad_buf(self, ru, shape):
        """[.., 2c] gradient buffer of a unit whose forward ran paired (this step), else None"""
        c0 = ru["units"][0][0]
        sv = self._saved.get(ru["prefix"])
        if not self.pair_bwd or c0.pair_with is None or sv is None or not sv.get("tpair"):
            return None
        return self._buf(f"{ru['prefix']}.mg", tuple(shape[:4]) + (2 * c0.cout,))

    def _dout_slot(self, ru, name, shape):
        """where the gradient of a residual unit's OUTPUT is to be written: the second half of the unit's pair
        buffer when its two stride-2 input gradients run as one launch, else the plain buffer `name`"""
        mg = self._pair_grad_buf(ru, shape)
        if mg is not None and shape[4] == ru["units"][0][0].cout:
            return mg[..., shape[4]:]
        return self._buf(name, shape)

    def _train_pair(self, ru, x, oshape):
        """the [.., 2c] buffer of the merged first subunit + residual convolution (training), or None"""
        c0 = ru["units"][0][0]
        if not self.pair_train or c0.pair_with is None or c0 in self._carry_convs:
            return None
        m = self._buf(f"{ru['prefix']}.tm", oshape + (2 * c0.cout,))
        return m if ops.conv3d_split_act_ok(x, m, 3, c0.stride) else None

    def _merged_eval(self, ru, x, oshape):
        """(pack, bias, buffer) of the merged subunit-0 + residual convolution of a unit (inference,
        MFMA layers, segmi_conv3d_fwd_split_act), or None.  The pack concatenates the BatchNorm-folded
        subunit-0 weight and the residual weight along the output channels; cached per weight version."""
        units, rc = ru["units"], ru["res"]
        if not self.merge_eval_pairs or rc is None or len(units) < 2 or units[0][1] is None:
            return None
        c0, bn0 = units[0]
        if not (c0.mfma and rc.mfma and c0.k == 3 and rc.k == 3 and c0.stride == rc.stride
                and c0.cout == rc.cout and c0.cin == rc.cin and not c0.transposed):
            return None
        m = self._buf(f"{ru['prefix']}.merged", oshape + (2 * c0.cout,))
        if not ops.conv3d_split_act_ok(x, m, 3, c0.stride):
            return None
        hit = ru.get("_merged")
        if hit is None or hit[0] != self.weights_version:
            sc, sh = bn0.eval_affine()
            w_cat = torch.cat([c0.w, rc.w], 0).contiguous()
            scale = torch.cat([sc, torch.ones_like(sc)])
            bias = torch.cat([torch.addcmul(sh, c0.b, sc), rc.b]).contiguous()
            pack = ops.wpack(self.dtype, 0, w_cat, c0.cin, 2 * c0.cout, 3, scale=scale)
            hit = (self.weights_version, pack, bias)
            ru["_merged"] = hit
        return hit[1], hit[2], m

    def _conv_train(self, conv: _Conv, x, y, bn: Optional[_BN], in_tf=None):
        """raw conv (+bias) with fused statistics, then finalize into bn.*"""
        stats = None
        rows = 0
        if bn is not None:
            rows = (ops.convT3d_stats_rows(x, y) if conv.transposed
                    else ops.conv3d_stats_rows(x, y, conv.k, conv.stride))
            stats = self._fstat(rows, conv.cout)
        # the launch that writes the statistics rows finalises them itself (segmi_bn_fin): no
        # bn_finalize launch on the dependent chain (SEGMI_FUSE_FIN=0: separate launch, for A/B)
        fin = self._stats_fin(bn, y) if bn is not None and self.fuse_fin else None
        if conv.transposed:
            self._timed(conv.prefix + ":fwd", ops.convT3d_fwd, x, y, conv.fwd_pack(), conv.w,
                        conv.b, stats=stats, stats_fin=fin)
        else:
            self._timed(conv.prefix + ":fwd", ops.conv3d_fwd, x, y, conv.fwd_pack(), conv.w, 0,
                        conv.b, conv.k, conv.stride, stats=stats, in_tf=in_tf, stats_fin=fin)
        if bn is not None and fin is None:
            count = y.shape[0] * y.shape[1] * y.shape[2] * y.shape[3]
            ops.bn_finalize(stats, rows, conv.cout, count, bn.gamma, bn.beta, bn.rm, bn.rv,
                            self.momentum, self.eps, bn.mean, bn.invstd, bn.scale, bn.shift)

    def _stats_fin(self, bn: "_BN", y) -> tuple:
        count = y.shape[0] * y.shape[1] * y.shape[2] * y.shape[3]
        return (count, bn.gamma, bn.beta, bn.rm, bn.rv, self.momentum, self.eps, bn.mean, bn.invstd,
                bn.scale, bn.shift)

    def _wgrad(self, conv: _Conv, x, dy, need_bias: bool = True, in_tf=None):
        """dW (and db) of `conv` given its forward input x and output gradient dy.

        ``need_bias=False`` for a conv that feeds a training-mode BatchNorm: dy is then the BN
        input gradient whose per-channel sum is identically zero (sum_v dr = gamma*invstd*(S -
        N*S/N - (T/N)*sum xhat) = 0), so db == 0 exactly; the arena entry stays at its initial
        zero instead of spending a pass over dy to compute rounding noise."""
        # Weight gradients have no consumer until the optimiser: they run on a side HIP stream,
        # concurrently with the dgrad / BatchNorm-backward chain of the main stream (the
        # mid / deep levels do not fill 256 CUs with one kernel at a time).  All wgrads share the
        # side stream, hence also their scratch buffer, in issue order.
        if self._diag_skip_wgrad:      # diagnostics only: time the main chain without its side-stream partner
            return
        if self._carry_open and conv in self._carry_convs:
            self._carried.append((conv, x, dy, need_bias, in_tf))     # issued after the end of backward
            return
        if self._defer_open:
            # issued later, when the main chain is down in the small deep levels (see defer_top_wgrad)
            self._deferred.append((conv, x, dy, need_bias, in_tf))
            return
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap_wgrad else None
        if side is not None:
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
        # beside the main chain the kernels size their grids for part of the chip (wgrad_cus_overlap); a per-call
        # argument of the C-ABI, so engines with different schedules do not interfere
        cus = self.wgrad_cus_overlap if side is not None else 0
        with torch.cuda.stream(side) if side is not None else _NullCtx():
            if conv.transposed:
                # dW_T[ci][co][tap]: stride-2 conv wgrad with x := dy (fine), dy := x (coarse)
                nbytes = ops.conv3d_wgrad_workspace(dy, x, 3, 2, cus)
                ws = self._scratch_buf("wgrad", nbytes)
                self._timed(conv.prefix + ":wgrad", ops.conv3d_wgrad, dy, x, conv.gw, None, 3, 2, ws, cus=cus)
                if need_bias:
                    ops.bias_grad(dy, conv.gb, ws)
            else:
                nbytes = ops.conv3d_wgrad_workspace(x, dy, conv.k, conv.stride, cus)
                ws = self._scratch_buf("wgrad", nbytes)
                self._timed(conv.prefix + ":wgrad", ops.conv3d_wgrad, x, dy, conv.gw,
                            conv.gb if need_bias else None, conv.k, conv.stride, ws, in_tf=in_tf, cus=cus)

    def _bsum_ok(self, conv: _Conv, dy, dx, bn: Optional["_BN"]) -> bool:
        """the BatchNorm-backward reduction of `bn` (whose output gradient is dx = dgrad(conv, dy)) can
        run in that input-gradient launch's epilogue (segmi_bn_bwd_sums)"""
        # 32 -> 32 layers (BASELINE config 4): the kernel takes the sums (tested), this engine does not ask for them --
        # the variant that fits its registers computes the 32 outputs as two 16-channel tiles on grid.y and the
        # 160^3 / batch 4 step went 9.75 -> 10.2 ms with it (both tiles in one workgroup: 96 spilled registers,
        # 10.9 ms).  SEGMI_BSUM32=1 asks for them anyway (A/B).
        return (bn is not None and self.fuse_bn_bwd and self.dtype == torch.bfloat16 and self.dropout_p <= 0.0
                and not conv.transposed and conv.stride == 1 and conv.k == 3 and conv.mfma
                and (conv.cout == 16 or self._bsum32)
                and ops.conv3d_bn_bwd_sums_ok(dy, dx, 3, 1))

    _bsum32 = os.environ.get("SEGMI_BSUM32", "0") == "1"

    def _dgrad(self, conv: _Conv, dy, dx, residual=None, bsum=None):
        """dx = dgrad(conv, dy) (+ residual).  ``bsum`` = (bn, x_raw): also the partial rows of that
        BatchNorm's backward reduction over (dx, x_raw); returns their row count (else 0)."""
        if bsum is not None:
            bn, x_raw = bsum
            rows = ops.conv3d_stats_rows(dy, dx, conv.k, 1)
            part = self._fstat(rows, bn.c)
            self._timed(conv.prefix + ":dgrad", ops.conv3d_fwd, dy, dx, conv.dgrad_pack(), conv.w, 1,
                        None, conv.k, 1, residual=residual,
                        bn_bwd=(x_raw, bn.mean, bn.invstd, bn.gamma, bn.beta, bn.alpha, part),
                        bn_bwd_fin=self._bwd_fin(bn, x_raw) if self.fuse_fin else None)
            return rows
        if conv.transposed:
            ops.conv3d_fwd(dy, dx, conv.dgrad_pack(), conv.w, 0, None, 3, 2, residual=residual)
        elif conv.stride == 2:
            ops.convT3d_fwd(dy, dx, conv.dgrad_pack(), conv.w, None, residual=residual)
        else:
            self._timed(conv.prefix + ":dgrad", ops.conv3d_fwd, dy, dx, conv.dgrad_pack(), conv.w, 1,
                        None, conv.k, 1, residual=residual)

    @staticmethod
    def _bwd_fin(bn: "_BN", x_raw) -> tuple:
        count = x_raw.shape[0] * x_raw.shape[1] * x_raw.shape[2] * x_raw.shape[3]
        return (count, bn.g_gamma, bn.g_beta, bn.g_alpha, bn.coef)

    # decoder levels: the BatchNorm-backward apply of the up layer rides in the stride-2 convolution that
    # consumes it (the transposed convolution's input gradient; csrc/conv_bnbwd_impl.h): du is written once
    # for the weight gradient instead of written + read.  SEGMI_FUSE_APPLY_CONV=0: two launches (A/B).
    fuse_apply_conv = os.environ.get("SEGMI_FUSE_APPLY_CONV", "1") != "0"

    def _bn_bwd(self, bn: _BN, dy, x_raw, dx, sums_rows: int = 0, then_conv=None) -> bool:
        """``sums_rows`` > 0: the reduction's partial rows were written by the launch that produced dy
        (``_dgrad(..., bsum=)``); only finalisation and apply remain.  ``then_conv`` = (conv, out): the
        caller's next launch is ``out = dgrad(conv, dx)`` of a transposed convolution; returns True when
        that convolution was done here, in the launch that computed dx."""
        count = x_raw.shape[0] * x_raw.shape[1] * x_raw.shape[2] * x_raw.shape[3]
        if (not sums_rows and self.fuse_bn_bwd_small and self.fuse_fin and self.dropout_p <= 0.0
                and ops.bn_act_bwd_fused_ok(dy, x_raw, dx)):
            # deep levels: reduce + finalise + apply as ONE launch (csrc/norm_act.hip, bn_act_bwd_fused_kernel)
            # its workgroups hold a CU each and wait on it for the launch's last one: beside the CU-exclusive
            # weight-gradient kernels of the second stream the grid is capped at the CUs those leave free, so no
            # workgroup of this launch waits for one of them to retire (fused_bn_max_wgs)
            rows = ops.bn_act_bwd_fused_rows(x_raw)
            ops.bn_act_bwd_fused(dy, x_raw, dx, bn.mean, bn.invstd, bn.gamma, bn.beta, bn.alpha,
                                 self._fstat(rows, bn.c), self._bwd_fin(bn, x_raw), max_wgs=self.fused_bn_max_wgs())
            return False
        if sums_rows:
            rows = sums_rows
            part = self._fstat(rows, bn.c)
        else:
            rows = ops.bn_act_bwd_rows(x_raw)
            part = self._fstat(rows, bn.c)
            ops.bn_act_bwd_reduce(dy, x_raw, bn.mean, bn.invstd, bn.gamma, bn.beta, bn.alpha, part,
                                  dropout=bn.drop(), fin=self._bwd_fin(bn, x_raw) if self.fuse_fin else None)
        if not self.fuse_fin:         # else: finalised by the launch that wrote the rows
            ops.bn_act_bwd_finalize(part, rows, bn.c, count, bn.gamma, bn.invstd, bn.g_gamma,
                                    bn.g_beta, bn.g_alpha, bn.coef)
        if (then_conv is not None and self.fuse_apply_conv and self.dropout_p <= 0.0 and then_conv[0].transposed
                and ops.bn_act_bwd_apply_conv_ok(dy, x_raw, dx, then_conv[1])):
            ops.bn_act_bwd_apply_conv(dy, x_raw, dx, bn.mean, bn.invstd, bn.gamma, bn.beta, bn.alpha, bn.coef,
                                      then_conv[1], then_conv[0].dgrad_pack())
            return True
        ops.bn_act_bwd_apply(dy, x_raw, dx, bn.mean, bn.invstd, bn.gamma, bn.beta, bn.alpha,
                             bn.coef, dropout=bn.drop())
        return False

    # ------------------------------------------------------------------ residual unit
    def _ru_fwd_train(self, ru, x, out, in_tf=None):
        """``in_tf``: x is the RAW output of the producer conv and (scale, shift, alpha) its pending
        BatchNorm-apply + PReLU, which the first unit's conv applies on the fly (only for a unit
        whose sole consumers of x are that conv, its weight gradient and an in-kernel identity
        residual -- see ``_level_fwd``)."""
        pre = ru["prefix"]
        n, d, h, w = self._down_shape(x.shape, ru["stride"])
        cur = x
        nun = len(ru["units"])
        saved = {"x": x}
        # residual branch straight into `out` (read back as the epilogue residual of the last
        # unit); it is independent of the conv-unit chain, so it runs on a side stream
        br = None
        paired = False
        tpair = None
        if ru["res"] is not None:
            rc = ru["res"]
            paired = self._pair_ok(ru, x, out, f"{pre}.r0", (n, d, h, w))
            if not paired and in_tf is None:
                tpair = self._train_pair(ru, x, (n, d, h, w))
            if not paired and tpair is None:
                br = self._fork_branch()
                with torch.cuda.stream(br) if br is not None else _NullCtx():
                    ops.conv3d_fwd(x, out, rc.fwd_pack(), rc.w, 0, rc.b, rc.k, rc.stride)
            resid = out
        else:
            resid = x
        for i, (conv, bn) in enumerate(ru["units"]):
            last = i == nun - 1
            if i == 0 and tpair is not None:
                # MFMA layers: subunit 0 (+ its statistics) and the residual convolution as ONE launch with 2c
                # outputs -- one staging of x instead of two; consumers read the halves as channel-slice views
                m = tpair
                r, resid = m[..., :conv.cout], m[..., conv.cout:]
                rows = ops.conv3d_stats_rows(x, m, conv.k, conv.stride)
                stats = self._fstat(rows, conv.cout)
                fin = self._stats_fin(bn, r) if self.fuse_fin else None
                self._timed(conv.prefix + ":fwd", ops.conv3d_fwd_split_act, x, m, conv.pair_pack(), conv.b, None,
                            conv.cout, conv.k, conv.stride, bias_b=ru["res"].b, stats=stats, stats_fin=fin)
                if fin is None:
                    ops.bn_finalize(stats, rows, conv.cout, n * d * h * w, bn.gamma, bn.beta, bn.rm,
                                    bn.rv, self.momentum, self.eps, bn.mean, bn.invstd, bn.scale, bn.shift)
                saved[f"in{i}"] = cur
                saved[f"r{i}"] = r
                nconv = ru["units"][i + 1][0]
                if self._tf_ok(r, r, nconv) and nconv.cin == conv.cout and nconv.cout == conv.cout:
                    in_tf = (bn.scale, bn.shift, bn.alpha)
                    cur = r
                else:
                    a = self._buf(f"{pre}.a{i}", (n, d, h, w, conv.cout))
                    ops.bn_act_fwd(r, a, bn.scale, bn.shift, bn.alpha, dropout=bn.drop())
                    cur = a
                continue
            if i == 0 and paired:
                # first layer: subunit 0 and the residual convolution share one staging of x
                r = self._buf(f"{pre}.r{i}", (n, d, h, w, conv.cout))
                rows = ops.conv3d_stats_rows(x, r, conv.k, conv.stride)
                stats = self._fstat(rows, conv.cout)
                rc = ru["res"]
                fin = self._stats_fin(bn, r) if self.fuse_fin else None
                self._timed(conv.prefix + ":fwd", ops.conv3d_fwd_pair, x, r, conv.w, conv.b, out,
                            rc.w, rc.b, conv.stride, stats_a=stats, stats_fin_a=fin)
                if fin is None:
                    ops.bn_finalize(stats, rows, conv.cout, n * d * h * w, bn.gamma, bn.beta, bn.rm,
                                    bn.rv, self.momentum, self.eps, bn.mean, bn.invstd, bn.scale, bn.shift)
                saved[f"in{i}"] = cur
                saved[f"r{i}"] = r
                saved["tpair"] = True
                nconv = ru["units"][i + 1][0]
                if self._tf_ok(r, r, nconv) and nconv.cin == conv.cout and nconv.cout == conv.cout:
                    in_tf = (bn.scale, bn.shift, bn.alpha)
                    cur = r
                else:
                    a = self._buf(f"{pre}.a{i}", (n, d, h, w, conv.cout))
                    ops.bn_act_fwd(r, a, bn.scale, bn.shift, bn.alpha, dropout=bn.drop())
                    cur = a
                continue
            if bn is None:
                # conv-only last unit (top of the net): out = conv(cur) + residual
                self._join_branch(br)
                self._timed(conv.prefix + ":fwd", ops.conv3d_fwd, cur, out, conv.fwd_pack(), conv.w,
                            0, conv.b, conv.k, conv.stride, residual=resid, in_tf=in_tf)
                saved[f"in{i}"] = cur
                saved[f"tf{i}"] = in_tf
                break
            r = self._buf(f"{pre}.r{i}", (n, d, h, w, conv.cout))
            self._conv_train(conv, cur, r, bn, in_tf=in_tf)
            saved[f"in{i}"] = cur
            saved[f"r{i}"] = r
            saved[f"tf{i}"] = in_tf
            in_tf = None
            if last:
                self._join_branch(br)
                ops.bn_act_fwd(r, out, bn.scale, bn.shift, bn.alpha, residual=resid, dropout=bn.drop())
            else:
                nconv = ru["units"][i + 1][0]
                if self._tf_ok(r, r, nconv) and nconv.cin == conv.cout and nconv.cout == conv.cout:
                    # the next unit's conv (and its weight gradient) normalise r on the fly
                    in_tf = (bn.scale, bn.shift, bn.alpha)
                    cur = r
                else:
                    a = self._buf(f"{pre}.a{i}", (n, d, h, w, conv.cout))
                    ops.bn_act_fwd(r, a, bn.scale, bn.shift, bn.alpha, dropout=bn.drop())
                    cur = a
        self._saved[pre] = saved

    def _ru_bwd(self, ru, dout, dx=None, extra=None, dx_bn=None):
        """Backward of a residual unit.  dout: grad of its output.  If dx is given it receives
        the input gradient (+ `extra`, an additional gradient flowing into the same input).
        ``dx_bn`` = (bn, x_raw): dx is the output gradient of that BatchNorm; when the unit's last
        launch into dx can carry the reduction of its backward, the partial-row count is returned
        (else 0)."""
        pre = ru["prefix"]
        sv = self._saved[pre]
        x = sv["x"]
        nun = len(ru["units"])
        rc = ru["res"]
        # paired input gradients: dout is the second half of the unit's [.., 2c] gradient buffer (the caller got it
        # from _dout_slot), subunit 0's BatchNorm backward writes the first half, ONE transposed convolution over the
        # buffer replaces the two input-gradient launches
        mg = self._pair_grad_buf(ru, dout.shape) if (rc is not None and dx is not None) else None
        if mg is not None and dout.data_ptr() != mg[..., dout.shape[4]:].data_ptr():
            mg = None
        # the residual conv's input gradient depends only on dout: side stream, joined before
        # the first unit's dgrad accumulates on top of it
        br = None
        if rc is not None and dx is not None and mg is None:
            br = self._fork_branch()
            with torch.cuda.stream(br) if br is not None else _NullCtx():
                self._dgrad(rc, dout, dx, residual=extra)
        g = dout            # gradient flowing back through the conv branch
        g_rows = 0          # > 0: the launch that wrote g also wrote the next BatchNorm's reduction rows
        for i in range(nun - 1, -1, -1):
            conv, bn = ru["units"][i]
            xin = sv[f"in{i}"]
            if bn is not None:
                r = sv[f"r{i}"]
                dr = mg[..., :r.shape[4]] if (mg is not None and i == 0) else self._buf(f"{pre}.dr{i}", r.shape)
                self._bn_bwd(bn, g, r, dr, sums_rows=g_rows)
            else:
                dr = g
            g_rows = 0
            self._wgrad(conv, xin, dr, need_bias=bn is None and conv is not self._top_bias_conv,
                        in_tf=sv.get(f"tf{i}"))
            if i > 0:
                da = self._buf(f"{pre}.da{i - 1}", xin.shape)
                pbn = ru["units"][i - 1][1]
                if self._bsum_ok(conv, dr, da, pbn):
                    g_rows = self._dgrad(conv, dr, da, bsum=(pbn, sv[f"r{i - 1}"]))
                else:
                    self._dgrad(conv, dr, da)
                g = da
            else:
                first_dr = dr
        if rc is not None:
            self._wgrad(rc, x, dout)
        # every parameter gradient of this unit (and of everything after it in the arena) is final
        self._grads_ready(self.param_offsets[f"model.{pre}.conv.unit0.conv.weight"][0])
        if dx is None:
            return 0
        conv0 = ru["units"][0][0]
        if mg is not None:
            ops.convT3d_fwd(mg, dx, conv0.pair_dgrad_pack(), None, None, residual=extra)
            return 0
        bsum = dx_bn if dx_bn is not None and self._bsum_ok(conv0, first_dr, dx, dx_bn[0]) else None
        if rc is not None:
            self._join_branch(br)
            return self._dgrad(conv0, first_dr, dx, residual=dx, bsum=bsum) or 0
        # identity residual: dx = dgrad(conv0) + dout (+ extra)
        if extra is not None:
            ops.add(dout, extra, dx)
            return self._dgrad(conv0, first_dr, dx, residual=dx, bsum=bsum) or 0
        return self._dgrad(conv0, first_dr, dx, residual=dout, bsum=bsum) or 0

    def _ru_fwd_eval(self, ru, x, out):
        pre = ru["prefix"]
        n, d, h, w = self._down_shape(x.shape, ru["stride"])
        br = None
        paired = False
        merged = None
        if ru["res"] is not None:
            rc = ru["res"]
            paired = self._pair_ok(ru, x, out, f"{pre}.ea0", (n, d, h, w))
            if isinstance(x, ops.WindowBatch) and not paired:
                raise RuntimeError("window views need the first-layer pair kernel (window_views_ok said yes?)")
            merged = None if paired else self._merged_eval(ru, x, (n, d, h, w))
            if not paired and merged is None:
                br = self._fork_branch()
                with torch.cuda.stream(br) if br is not None else _NullCtx():
                    ops.conv3d_fwd(x, out, rc.fwd_pack(), rc.w, 0, rc.b, rc.k, rc.stride)
            resid = out
        else:
            resid = x
        cur = x
        nun = len(ru["units"])
        for i, (conv, bn) in enumerate(ru["units"]):
            last = i == nun - 1
            if i == 0 and merged is not None:
                # subunit 0 and the residual convolution as ONE launch with 2c outputs (one staging of
                # x); consumers read the halves as channel-slice views
                pack, bias, m = merged
                ops.conv3d_fwd_split_act(x, m, pack, bias, bn.alpha, conv.cout, 3, conv.stride)
                cur, resid = m[..., :conv.cout], m[..., conv.cout:]
                continue
            if i == 0 and paired:
                sc, sh = bn.eval_affine()
                _, wsrc, bias = conv.folded(sc, sh)
                dst = self._buf(f"{pre}.ea{i}", (n, d, h, w, conv.cout))
                rc = ru["res"]
                ops.conv3d_fwd_pair(x, dst, wsrc, bias, out, rc.w, rc.b, conv.stride,
                                    prelu_alpha_a=bn.alpha)
                cur = dst
                continue
            if bn is None:
                self._join_branch(br)
                self._timed(conv.prefix + ":fwd", ops.conv3d_fwd, cur, out, conv.fwd_pack(), conv.w, 0,
                            conv.b, conv.k, conv.stride, residual=resid)
                break
            sc, sh = bn.eval_affine()
            pack, wsrc, bias = conv.folded(sc, sh)
            dst = out if last else self._buf(f"{pre}.ea{i}", (n, d, h, w, conv.cout))
            if last:
                self._join_branch(br)
            ops.conv3d_fwd(cur, dst, pack, wsrc, 0, bias, conv.k, conv.stride,
                           prelu_alpha=bn.alpha, residual=resid if last else None)
            cur = dst

    # ------------------------------------------------------------------ levels
    def _level_fwd(self, lvl, x, out, train: bool):
        p = lvl["prefix"]
        n, d, h, w = self._down_shape(x.shape, lvl["stride"])
        c, upc, subc = lvl["c"], lvl["upc"], lvl["subc"]
        tag = "t" if train else "e"
        if (train and id(lvl) in self._carry_lvls and self.carry_top_wgrad and self.grad_hook is None
                and self.overlap_wgrad):
            # the carried transposed-conv weight gradient of the LAST step may still be reading the last
            # step's skip buffer: alternate between two
            tag = "t" + str(self._fwd_parity)
        cat = self._buf(f"{p}cat.{tag}", (n, d, h, w, upc))
        down_out = cat[..., :c]
        sub_out = cat[..., c:]
        ru_fwd = self._ru_fwd_train if train else self._ru_fwd_eval
        ru_fwd(lvl["down"], x, down_out)
        if lvl["sub"] is not None:
            self._level_fwd(lvl["sub"], down_out, sub_out, train)
        else:
            ru_fwd(lvl["bottom"], down_out, sub_out)
        up, ubn = lvl["upconv"], lvl["upbn"]
        oshape = (x.shape[0], x.shape[1], x.shape[2], x.shape[3], lvl["outc"])
        if train:
            if id(lvl) in self._carry_lvls:
                self.sync_weights()        # parameters of the carried layers, and their readers of u / scale / shift
            u = self._buf(f"{p}u", oshape)
            self._conv_train(up, cat, u, ubn)
            upru = lvl["upru"]
            conv0, bn0 = upru["units"][0]
            if (bn0 is None and upru["res"] is None and conv0.cin == conv0.cout == lvl["outc"]
                    and self._tf_ok(u, out, conv0)):
                # conv-only unit with an identity residual (the top of the net): its conv, the weight
                # gradient and the in-kernel residual are the only readers of act(bn(u)) -- they
                # apply it themselves, the normalised tensor is never written
                self._saved[p + "up"] = {"cat": cat, "u": u}
                self._ru_fwd_train(upru, u, out, in_tf=(ubn.scale, ubn.shift, ubn.alpha))
            else:
                au = self._buf(f"{p}au.t", oshape)
                ops.bn_act_fwd(u, au, ubn.scale, ubn.shift, ubn.alpha, dropout=ubn.drop())
                self._saved[p + "up"] = {"cat": cat, "u": u}
                self._ru_fwd_train(upru, au, out)
        else:
            sc, sh = ubn.eval_affine()
            upru = lvl["upru"]
            conv0, bn0 = upru["units"][0]
            if (self.fuse_eval_top and lvl["is_top"] and self.dtype == torch.bfloat16 and bn0 is None
                    and upru["res"] is None and up.cin == 32 and up.cout == 16 and conv0.cin == conv0.cout == 16
                    and ops.dectop_ok(cat, out)):
                # the full-resolution decoder as ONE launch: the 16-channel tensor between the
                # transposed conv and the conv-only unit never reaches HBM (csrc/dectop.hip)
                hit = lvl.get("_dectop")
                if hit is None or hit[0] != self.weights_version:
                    # (one host read of the slope per weights version: the kernel's PReLU fast path)
                    a = float(ubn.alpha.reshape(-1)[0])
                    hit = (self.weights_version, ops.dectop_up_frag(up.w, sc), torch.addcmul(sh, up.b, sc),
                           0.0 <= a <= 1.0)
                    lvl["_dectop"] = hit
                self._timed(conv0.prefix + ":fwd", ops.dectop_fwd, cat, out, hit[1], hit[2], ubn.alpha,
                            conv0.fwd_pack(), conv0.b, alpha_in_unit_range=hit[3])
                self.eval_top_fused = True
                return
            au = self._buf(f"{p}au.e", oshape)
            pack, wsrc, bias = up.folded(sc, sh)
            ops.convT3d_fwd(cat, au, pack, wsrc, bias, prelu_alpha=ubn.alpha)
            self._ru_fwd_eval(lvl["upru"], au, out)

    def _level_bwd(self, lvl, dout, dx=None, extra=None):
        if self._defer_open and self._bwd_depth >= self.defer_flush_depth:
            self._flush_deferred()
        self._bwd_depth += 1
        try:
            self._level_bwd_body(lvl, dout, dx, extra)
        finally:
            self._bwd_depth -= 1

    def _level_bwd_body(self, lvl, dout, dx=None, extra=None):
        p = lvl["prefix"]
        sv = self._saved[p + "up"]
        cat, u = sv["cat"], sv["u"]
        c = lvl["c"]
        dau = self._buf(f"{p}dau", u.shape)
        rows = self._ru_bwd(lvl["upru"], dout, dx=dau, dx_bn=(lvl["upbn"], u))
        du = self._buf(f"{p}du", u.shape)
        up = lvl["upconv"]
        dcat = self._buf(f"{p}dcat", cat.shape)
        conv_done = self._bn_bwd(lvl["upbn"], dau, u, du, sums_rows=rows, then_conv=(up, dcat))
        self._wgrad(up, cat, du, need_bias=False)
        self._grads_ready(self.param_offsets[f"model.{p}2.0.conv.weight"][0])
        if not conv_done:
            self._dgrad(up, du, dcat)
        d_down, d_sub = dcat[..., :c], dcat[..., c:]
        dsum = self._dout_slot(lvl["down"], f"{p}ddown", (cat.shape[0], cat.shape[1], cat.shape[2], cat.shape[3], c))
        if lvl["sub"] is not None:
            self._level_bwd(lvl["sub"], d_sub, dx=dsum, extra=d_down)
        else:
            self._ru_bwd(lvl["bottom"], d_sub, dx=dsum, extra=d_down)
        self._ru_bwd(lvl["down"], dsum, dx=dx, extra=extra)

    # ------------------------------------------------------------------ public API
    def _prep_input(self, x: torch.Tensor) -> torch.Tensor:
        """x: [N, C, D, H, W] float32 (reference layout; [N, C, H, W] or depth 1 for a 2-D
        network) -> NDHWC compute-dtype tensor."""
        if isinstance(x, ops.WindowBatch):      # sliding windows read in place by the first-layer kernel
            if not self.window_views_ok(x.dtype):
                raise ValueError("this network / precision cannot read window views (ask window_views_ok)")
            total_stride = 1
            for s in self.net.strides[:len(self.net.channels) - 1]:
                total_stride *= s
            if any(r % total_stride for r in x.roi):
                raise ValueError(f"window extent {x.roi} is not divisible by the network's total stride {total_stride}")
            return x
        if self.net.dimensions == 2 and x.dim() == 4:
            x = x.unsqueeze(2)
        if x.dim() != 5 or x.shape[1] != self.net.in_channels or (self.net.dimensions == 2 and x.shape[2] != 1):
            raise ValueError(f"expected input [N,{self.net.in_channels},"
                             f"{'D,' if self.net.dimensions == 3 else ''}H,W], got {tuple(x.shape)}")
        total_stride = 1
        for s in self.net.strides[:len(self.net.channels) - 1]:
            total_stride *= s
        for s in x.shape[(3 if self.net.dimensions == 2 else 2):]:
            if s % total_stride != 0:
                # the reference fails here too (torch.cat of mismatching skip tensors)
                raise ValueError(f"spatial extent {s} is not divisible by the network's total "
                                 f"stride {total_stride}")
        x = x.to(self.device)
        n, c, d, h, w = x.shape
        xin = self._buf("input", (n, d, h, w, c))
        if c == 1 and x.dtype == self.dtype and x.is_contiguous():
            return x.view(n, d, h, w, 1)       # already NDHWC in the compute dtype (e.g. bf16 windows)
        if c == 1 and x.dtype == torch.float32 and x.is_contiguous():
            src = x.view(n, d, h, w, 1)
            if self.dtype == torch.float32:
                return src
            ops.cast_copy(src, xin)
            return xin
        ops.nchw_to_ndhwc(x.float().contiguous(), xin)
        return xin

    def window_views_ok(self, dtype) -> bool:
        """inference: the first residual unit can read its windows straight from the volume
        (``ops.WindowBatch`` -> ``segmi_windows``): single input channel, 3-D, the small-Cin pair kernel
        (subunit 0 + residual convolution in one launch) takes the unit, windows in the compute dtype"""
        ru = self.levels["down"]
        units, rc = ru["units"], ru["res"]
        if self.net.in_channels != 1 or self.net.dimensions != 3 or dtype != self.dtype:
            return False
        if len(units) < 2 or units[0][1] is None or rc is None:
            return False
        c0 = units[0][0]
        return (not c0.mfma and not rc.mfma and c0.k == 3 and rc.k == 3 and c0.stride == rc.stride
                and c0.cout == rc.cout and c0.cout in (16, 32))

    def forward(self, x: torch.Tensor, train: Optional[bool] = None,
                out: Optional[torch.Tensor] = None, lane: int = 0) -> torch.Tensor:
        """[N,C,D,H,W] f32 -> logits NDHWC tensor [N,D,H,W,K] (compute dtype).

        ``out`` (eval mode): caller-owned dense NDHWC destination of the compute dtype, e.g. a
        slot range of the sliding-window prediction cache; otherwise an engine buffer that the
        next call overwrites.  ``lane`` (eval mode): activation-buffer set to use, so that forwards
        issued on different streams can be in flight together (weights and folded packs are shared:
        the caller orders the first forward after a weight change before any concurrent one)."""
        train = self.training if train is None else train
        if lane and train:
            raise ValueError("lanes are an inference-only feature")
        self._lane = int(lane)
        try:
            return self._forward(x, train, out)
        finally:
            self._lane = 0

    _fwd_parity = 0

    def _forward(self, x, train, out):
        if not train:
            self.sync_weights()
        xin = self._prep_input(x)
        n, d, h, w, _ = xin.shape
        k = self.net.out_channels
        shape = (n, d, h, w, self.kpad)
        if out is not None:
            if train:
                raise ValueError("forward(out=...) is an inference-only path")
            if self.kpad != k:
                raise ValueError("forward(out=...) needs a class count that is a multiple of 16")
            if tuple(out.shape) != shape or out.dtype != self.dtype or not out.is_contiguous():
                raise ValueError(f"out must be a contiguous {shape} {self.dtype} tensor")
            logits = out
        else:
            logits = self._buf("logits." + ("t" if train else "e"), shape)
        if train:
            self._fwd_parity ^= 1
            self._repack_async()
            self._saved.clear()
            self._nbt_flat += 1        # every BatchNorm runs exactly once per training forward
            self._drop_step += 1       # fresh dropout masks for this step (shared by its backward)
        self._level_fwd(self.levels, xin, logits, train)
        return logits if self.kpad == k else logits[..., :k]     # the K real classes (view, ld = kpad)

    def dlogits_buffer(self, logits: torch.Tensor) -> torch.Tensor:
        """Gradient buffer matching ``forward``'s result: the first K channels of a zero-initialised
        kpad-channel tensor (a loss kernel writes the K real channels, the padding stays 0)."""
        n, d, h, w, k = logits.shape
        full = self._bufs.get("dlogits")
        if full is None or tuple(full.shape) != (n, d, h, w, self.kpad) or full.dtype != self.dtype:
            full = torch.zeros((n, d, h, w, self.kpad), dtype=self.dtype, device=self.device)
            self._bufs["dlogits"] = full
        return full if self.kpad == k else full[..., :k]

    def top_bias_grad(self) -> torch.Tensor:
        """Gradient slot of the bias of the conv that produces the logits (f32[K]): a loss kernel
        that already walks dlogits can fill it (``backward(..., top_bias_done=True)``)."""
        return self.levels["upru"]["units"][-1][0].gb

    def backward(self, dlogits: torch.Tensor, top_bias_done: bool = False, carry: bool = False) -> bool:
        """dlogits NDHWC (compute dtype).  Fills the flat gradient arena (overwrites).

        ``carry`` (``Net.training_step`` only): the weight gradients of the two full-resolution decoder
        convolutions are issued on the weight-gradient stream AFTER the main stream has joined it, i.e.
        they are not part of what the caller's stream waits for: the caller must finish the step with
        ``finish_carried()`` (optimiser update of the arena suffix ``[carry_lo, n)`` + re-pack on that
        stream) and update only ``[0, carry_lo)`` itself.  Returns True when that is the case."""
        if not self._saved:
            raise RuntimeError("backward() needs a preceding training-mode forward()")
        self._top_bias_conv = self.levels["upru"]["units"][-1][0] if top_bias_done else None
        if dlogits.shape[4] != self.kpad:       # K real classes -> the padded gradient tensor
            full = self._bufs.get("dlogits")
            own = (full is not None and full.data_ptr() == dlogits.data_ptr()
                   and tuple(full.shape[:4]) == tuple(dlogits.shape[:4]) and dlogits.stride(3) == self.kpad)
            if not own:
                full = self.dlogits_buffer(dlogits)
                full = self._bufs["dlogits"]
                full[..., :dlogits.shape[4]].copy_(dlogits)
            dlogits = full
        self._deferred = []
        self._defer_open = self.defer_top_wgrad and self.grad_hook is None and self.overlap_wgrad
        self._carried = []
        self._carry_open = (carry and self.carry_top_wgrad and self.grad_hook is None and self.overlap_wgrad
                            and not self._diag_skip_wgrad)
        self._bwd_depth = 0
        try:
            self._level_bwd(self.levels, dlogits)
            if self._defer_open:                 # a net shallower than the flush depth
                self._flush_deferred()
        finally:
            self._defer_open = False
            carried, self._carry_open = self._carry_open, False
        self._top_bias_conv = None
        self._join_side()
        if carried:
            # behind the join: nothing on the caller's stream waits for these
            todo, self._carried = self._carried, []
            for conv, x, dy, need_bias, in_tf in todo:
                self._wgrad(conv, x, dy, need_bias=need_bias, in_tf=in_tf)
        return carried

    # Single-GPU training_step: the weight gradients of the two full-resolution decoder convolutions
    # (0.73 ms of HBM-bound persistent kernels, 1.6 GB) are carried over the end of the step: beside the
    # backward's own bandwidth-bound kernels they only stretch it (backward 2.30 ms alone, 3.79 ms with
    # all weight gradients beside it, 4.13 ms serial: the overlap buys 0.34 ms), beside the NEXT step's
    # forward -- whose 64^3 .. 8^3 levels are latency-bound and leave the memory system idle -- they are
    # nearly free.  Their layers are the last ones a forward reaches, so there is ~1 ms of slack: the
    # weight-gradient stream runs them, then the optimiser on the arena suffix [carry_lo, n) and the
    # re-pack of the two layers, and records `_tail_ev`; the next forward waits for it in front of the
    # top transposed convolution.  Same arithmetic in the same order per parameter: bit-identical weights
    # (tests/test_e2e_gpu.py).  SEGMI_CARRY_TOP_WGRAD=0: everything inside the step, as before.
    carry_top_wgrad = os.environ.get("SEGMI_CARRY_TOP_WGRAD", "1") != "0"
    # how many of the upper levels' up paths are carried (1 = the full-resolution decoder only).  Round 4: 2 -- the
    # weight-gradient stream was still ~0.2 ms behind when the main chain reached the optimiser (the 64^3 decoder's
    # weight gradients sat in front of the first encoder layers', which the next forward needs first); with the
    # 64^3 up path carried too the step went 5.02 -> 4.91 ms (alternating runs); 3: 4.99, 4: 5.25.
    carry_levels = int(os.environ.get("SEGMI_CARRY_LEVELS", "2"))
    _carry_open = False
    _carried: list = []
    _tail_ev = None
    _packs_stream = None

    def _pack_stream(self):
        if self._packs_stream is None:
            self._packs_stream = streams.shared_stream(self.device, streams.PACK)
        return self._packs_stream

    def finish_carried(self, update_suffix, new_version: int):
        """``update_suffix()``: the optimiser kernel over ``[carry_lo, n)``.  Runs it and the re-pack of the
        carried layers on the weight-gradient stream behind their weight gradients."""
        main, side = torch.cuda.current_stream(), self._side_stream()
        ev = torch.cuda.Event()
        ev.record(main)              # the suffix's BatchNorm / bias gradients come from the main chain
        side.wait_event(ev)
        with torch.cuda.stream(side):
            update_suffix()
            self.repack_tail(new_version)
            self._tail_ev = torch.cuda.Event()
            self._tail_ev.record(side)

    def sync_weights(self):
        """the current stream waits until every parameter (and pack) of the last training step is final"""
        if self._tail_ev is not None:
            torch.cuda.current_stream().wait_event(self._tail_ev)

    _top_bias_conv = None

    # ------------------------------------------------------------------ live kernel timing
    # bench.py sets `timed = {"<conv prefix>:<fwd|wgrad|dgrad>"}`; the matching C-ABI call is
    # bracketed by HIP events on the launch stream (torch's current stream).
    timed: Optional[set] = None

    def _timed(self, key: str, fn, *a, **k):
        if not self.timed or key not in self.timed:
            return fn(*a, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*a, **k)
        e.record()
        self.timings.setdefault(key, []).append((s, e))
        return out

    def timing_ms(self, key: str) -> List[float]:
        return [s.elapsed_time(e) for s, e in self.timings.get(key, [])]

    # callable(lo_offset, events): gradients at arena offsets >= lo are final once the main stream's
    # current point and `events` (side-stream weight gradients) have been reached
    grad_hook = None
    # SEGMI_SERIAL=1 keeps every kernel on one stream (per-kernel profiling without co-running work)
    overlap_wgrad = os.environ.get("SEGMI_SERIAL", "0") != "1"
    _side = None
    # CUs the weight-gradient kernels size their grids for while they run on the side stream
    # (the `cus` argument of segmi_conv3d_wgrad; SEGMI_WGRAD_CUS overrides).  Their workgroups hold a CU exclusively (768 threads
    # or ~400 registers each): one per CU on all 256 and the main chain's kernels wait for a CU to retire --
    # "overlap" was then mostly alternation.  Sized for half the chip both streams run: 5.55 -> 5.28 ms per
    # step (160: 5.40, 96: 5.53, 64: 5.93; `gpurun_out/r3/wcus_ab2.txt`).  Round 2 measured the same knob
    # neutral; what changed is that the deferred / carried schedule now puts these kernels beside the
    # latency-bound levels of the main chain, which need CUs, not bandwidth.
    wgrad_cus_overlap = 128

    # SEGMI_SIDE_CUS=k: the weight-gradient stream may use only k CUs (k / 8 per XCD; a CU-masked HIP
    # stream) -- set SEGMI_WGRAD_CUS to the same value so the persistent kernels size their grids for it
    _side_cus = int(os.environ.get("SEGMI_SIDE_CUS", "0") or 0)

    def _side_stream(self):
        if self._side is None:
            if self._side_cus:
                self._side = ops.cu_masked_stream(self._side_cus, self.device)
            else:
                self._side = streams.shared_stream(self.device, streams.WGRAD)
        return self._side

    # residual-branch overlap on a second side stream: measured neutral on MI355X (the branch
    # convs are short and the join sits on the critical path), so it is off by default
    overlap_branches = os.environ.get("SEGMI_OVERLAP_BRANCHES", "0") == "1"
    # BatchNorm-apply + PReLU folded into the consumer conv's staging where the kernels allow it
    # (segmi_in_affine); SEGMI_FUSE_BN=0 keeps the separate pass for A/B measurements
    fuse_bn_apply = os.environ.get("SEGMI_FUSE_BN", "1") != "0"
    # Finalisation of every BatchNorm reduction (forward statistics, backward sums) by the launch that
    # writes the partial rows (csrc/fin_tail.h); SEGMI_FUSE_FIN=0 keeps the separate one-workgroup launches
    fuse_fin = os.environ.get("SEGMI_FUSE_FIN", "1") != "0"
    # BatchNorm / PReLU backward of the small (<= 32 MB) tensors as one launch with a grid-wide hand-off
    # instead of reduce -> apply (SEGMI_FUSE_BN_BWD_SMALL=0: two launches)
    fuse_bn_bwd_small = os.environ.get("SEGMI_FUSE_BN_BWD_SMALL", "1") != "0"
    _fused_wgs_env = int(os.environ.get("SEGMI_FUSED_BN_WGS", "-1"))      # A/B: -1 = derive from the schedule

    def fused_bn_max_wgs(self) -> int:
        """workgroups the one-launch BatchNorm backward may hold: the CUs the weight-gradient stream's budget
        leaves free (0 = the device's capacity when nothing CU-exclusive runs beside it)"""
        if self._fused_wgs_env >= 0:
            return self._fused_wgs_env
        return max(8, 256 - ops.wgrad_cus(self.wgrad_cus_overlap)) if self.overlap_wgrad else 0
    # BatchNorm-backward reduction in the epilogue of the input-gradient launch that produces its
    # operand (segmi_bn_bwd_sums); SEGMI_FUSE_BN_BWD=0 keeps the separate two-tensor pass (A/B)
    fuse_bn_bwd = os.environ.get("SEGMI_FUSE_BN_BWD", "1") != "0"
    # Single-GPU training: the weight gradients of the decoder levels above the deepest one (among them
    # the persistent wgrad_ws launches of the two full-resolution levels, ~0.7 ms) are ISSUED only when
    # the main chain enters the deepest level: beside the big bandwidth-bound input-gradient /
    # BatchNorm-backward kernels of the upper levels they only take CUs and fabric away, beside the
    # latency-bound 16^3 / 8^3 kernels they fill an idle chip.  5.64-5.72 vs 5.69-5.77 ms per step
    # (alternating runs, one box; flushing one level earlier: no gain, at the very end: 6.2 ms).
    # Not with a grad_hook (data parallel): the arena suffix would become final later and every gradient
    # bucket with it (tried with the notifications held back until the flush: 5.89-5.94 vs 5.86-5.87 ms
    # per step with the buckets going through RCCL on one rank).  SEGMI_DEFER_DEPTH overrides the level at which
    # the queue is flushed.
    # OFF by default since the weight-gradient kernels are sized for half the chip (wgrad_cus_overlap): they no
    # longer take every CU away from the main chain, and the earlier they start the more of them hides --
    # 5.20-5.29 without against 5.32-5.40 ms with the deferral, three alternating runs on one box, 5.17-5.23
    # against 5.23-5.30 on another (`gpurun_out/r3/sched_ab.txt`, `sched2_ab.txt`).  SEGMI_DEFER_TOP_WGRAD=1: on.
    defer_top_wgrad = os.environ.get("SEGMI_DEFER_TOP_WGRAD", "0") != "0"
    _diag_skip_wgrad = os.environ.get("SEGMI_DIAG_SKIP_WGRAD") == "1"      # WRONG gradients: timing probes only
    _defer_depth_env = os.environ.get("SEGMI_DEFER_DEPTH")

    @property
    def defer_flush_depth(self) -> int:
        if self._defer_depth_env is not None:
            return int(self._defer_depth_env)
        depth, lvl = 0, self.levels
        while lvl.get("sub") is not None:
            depth, lvl = depth + 1, lvl["sub"]
        return depth                                  # index of the deepest level
    _defer_open = False
    _deferred: list = []
    _bwd_depth = 0

    def _flush_deferred(self):
        self._defer_open = False
        todo, self._deferred = self._deferred, []
        for conv, x, dy, need_bias, in_tf in todo:
            self._wgrad(conv, x, dy, need_bias=need_bias, in_tf=in_tf)
    _side2 = None

    def _fork_branch(self):
        """second side stream, ordered after the main stream's current point (None = disabled)"""
        if not self.overlap_branches:
            return None
        if self._side2 is None:
            self._side2 = streams.shared_stream(self.device, streams.AUX)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._side2.wait_event(ev)
        return self._side2

    @staticmethod
    def _join_branch(s):
        if s is not None:
            torch.cuda.current_stream().wait_stream(s)

    def _join_side(self):
        """main stream waits for every weight-gradient kernel issued so far"""
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def _grads_ready(self, lo: int):
        if self.grad_hook is not None:
            # the hook (bucketed all-reduce on its own stream) orders itself after the main stream's
            # current point AND after the weight-gradient stream -- the main stream itself does not
            # wait for the weight gradients here (that stall cost the overlap in data-parallel runs)
            after = []
            if self._side is not None:
                ev = torch.cuda.Event()
                ev.record(self._side)
                after.append(ev)
            self.grad_hook(lo, after)

    def bump(self):
        """Call after the parameters changed (optimiser step, load_state_dict)."""
        self.weights_version += 1
