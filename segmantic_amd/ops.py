"""Thin torch-tensor wrappers over the C-ABI of ``libsegmi.so``.

PyTorch is plumbing here (device memory, streams): every function below hands raw device
pointers + extents to a HIP kernel and returns.  Activations are ``[N, D, H, W, C]`` tensors
(NDHWC); a channel slice ``t[..., a:b]`` of such a tensor is a valid view (concat by offset).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import SEGMI_BF16, SEGMI_F32, Act, BnBwdFin, BnBwdSums, BnFin, InAffine, Windows, check, lib

_DT = {torch.float32: SEGMI_F32, torch.bfloat16: SEGMI_BF16}


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported activation dtype {t.dtype} (float32 / bfloat16 only)")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_device(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            "segmantic_amd ops run on an MI355X only (tensor is on %s); there is no CPU path"
            % t.device)


class WindowBatch:
    """A batch of sliding windows read IN PLACE: ``volume`` [D, H, W] (contiguous, single channel, compute
    dtype) and the (z, y, x) origins of <= 32 windows of extent ``roi`` that lie inside it.  Quacks like the
    NDHWC tensor [n, *roi, 1] it stands for (``shape``, ``dtype``, ``device``); only the first-layer pair
    kernel (``conv3d_fwd_pair``) reads it (``segmi_windows``)."""

    def __init__(self, volume: torch.Tensor, starts, roi):
        if volume.dim() != 3 or not volume.is_contiguous():
            raise ValueError("WindowBatch: volume must be a contiguous [D, H, W] tensor")
        self.volume, self.roi = volume, tuple(int(r) for r in roi)
        self.starts = [tuple(int(v) for v in s) for s in starts]
        D, H, W = volume.shape
        if not 1 <= len(self.starts) <= SW_MAX_VIEWS:
            raise ValueError(f"WindowBatch: 1 .. {SW_MAX_VIEWS} windows")
        for s in self.starts:
            if any(a < 0 or a + r > n for a, r, n in zip(s, self.roi, (D, H, W))):
                raise ValueError(f"WindowBatch: window {s} + {self.roi} leaves the volume {tuple(volume.shape)}")
        self.shape = (len(self.starts),) + self.roi + (1,)
        self.dtype, self.device, self.is_cuda = volume.dtype, volume.device, volume.is_cuda

    @staticmethod
    def eligible(volume_shape, starts, roi) -> bool:
        """4-element alignment of rows and window origins (the kernel's staging loads)"""
        return (volume_shape[2] % 4 == 0 and roi[2] % 4 == 0 and (volume_shape[1] * volume_shape[2]) % 4 == 0
                and all(s[2] % 4 == 0 for s in starts))

    def windows(self) -> Windows:
        D, H, W = self.volume.shape
        w = Windows()
        w.count, w.row_stride, w.plane_stride = len(self.starts), W, H * W
        for i, (z, y, x) in enumerate(self.starts):
            w.offset[i] = (z * H + y) * W + x
        return w


def act(t) -> Act:
    """NDHWC view descriptor of a 5-D tensor whose last dim has stride 1."""
    if isinstance(t, WindowBatch):
        n, d, h, w, _ = t.shape
        return Act(t.volume.data_ptr(), n, d, h, w, 1, 1)
    _require_device(t)
    if t.dim() != 5:
        raise ValueError(f"expected a 5-D NDHWC tensor, got shape {tuple(t.shape)}")
    n, d, h, w, c = t.shape
    sn, sd, sh, sw, sc = t.stride()
    ld = sw
    if c > 1 and sc != 1:
        raise ValueError("channel dim must be contiguous")
    if w == 1:
        ld = max(ld, c)
    if not (sh == w * ld or h == 1) or not (sd == h * w * ld or d == 1) or \
            not (sn == d * h * w * ld or n == 1):
        raise ValueError(f"tensor is not a dense NDHWC view: shape {tuple(t.shape)} "
                         f"stride {t.stride()}")
    return Act(t.data_ptr(), n, d, h, w, c, ld)


def _ref(a: Optional[Act]):
    return C.byref(a) if a is not None else None


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    _require_device(t)
    return C.c_void_p(t.data_ptr())


# ------------------------------------------------------------------ weights
def wpack_bytes(dtype: torch.dtype, kind: int, cin_k: int, cout_k: int, ksize: int) -> int:
    return int(lib.segmi_wpack_bytes(_DT[dtype], kind, cin_k, cout_k, ksize))


class WpackBatch:
    """Descriptor table for ``segmi_wpack_batch``: built once, re-run after every weight update.

    ``entries`` = [(kind, w_src, scale_or_None, cin_k, cout_k, ksize[, w_src2, cout_split]), ...];
    ``self.packed[i]`` is the device buffer entry ``i`` writes.  ``w_src2`` (kind 0): the source of the output
    channels from ``cout_split`` on (the pair pack of ``conv3d_fwd_split_act`` out of two parameter tensors)."""

    def __init__(self, dtype: torch.dtype, entries):
        self.dtype = dtype
        self.n = len(entries)
        self.packed = []
        self._keep = []
        self._host = (_lib.WpackDesc * self.n)()
        dev = entries[0][1].device
        for i, ent in enumerate(entries):
            kind, w, scale, cin_k, cout_k, ks = ent[:6]
            w2, split = (ent[6], ent[7]) if len(ent) > 6 else (None, 0)
            nbytes = wpack_bytes(dtype, kind, cin_k, cout_k, ks)
            if nbytes <= 0:
                raise ValueError(f"no MFMA pack for cin={cin_k} cout={cout_k}")
            out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self.packed.append(out)
            self._keep.append((w, scale, w2))
            d = self._host[i]
            _require_device(w)
            d.w_src, d.packed = w.data_ptr(), out.data_ptr()
            if w2 is not None:
                if not ((kind == 0 and 0 < split < cout_k) or (kind == 2 and 0 < split < cin_k)):
                    raise ValueError("WpackBatch: a second source needs kind 0 (0 < split < cout_k) or kind 2 "
                                     "(0 < split < cin_k)")
                _require_device(w2)
                d.w_src2, d.cout_split = w2.data_ptr(), int(split)
            d.scale = scale.data_ptr() if scale is not None else None
            d.kind, d.cin_k, d.cout_k, d.ksize = kind, cin_k, cout_k, ks
        self._dev = torch.empty(C.sizeof(_lib.WpackDesc) * self.n, dtype=torch.uint8, device=dev)
        self._uploaded = False

    def run(self):
        check(lib.segmi_wpack_batch(_DT[self.dtype], self._host, self.n, _ptr(self._dev),
                                    0 if self._uploaded else 1, _stream()), "wpack_batch")
        self._uploaded = True


def wpack(dtype: torch.dtype, kind: int, w_src: torch.Tensor, cin_k: int, cout_k: int,
          ksize: int, scale: Optional[torch.Tensor] = None,
          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    nbytes = wpack_bytes(dtype, kind, cin_k, cout_k, ksize)
    if nbytes <= 0:
        raise ValueError(f"no MFMA pack for cin={cin_k} cout={cout_k}")
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=w_src.device)
    check(lib.segmi_wpack(_DT[dtype], kind, _ptr(w_src), _ptr(scale), cin_k, cout_k, ksize,
                          _ptr(out), _stream()), "wpack")
    return out


def mfma_ok(cin: int, cout: int) -> bool:
    return cin % 16 == 0 and cout % 16 == 0


# ------------------------------------------------------------------ convolution
def conv3d_stats_rows(x, y, ksize, stride) -> int:
    ax, ay = act(x), act(y)
    return int(lib.segmi_conv3d_stats_rows(dtype_code(x), C.byref(ax), C.byref(ay), ksize,
                                           stride))


def _in_affine(in_tf):
    """(scale, shift, prelu_alpha | None) device tensors -> segmi_in_affine (or NULL)"""
    if in_tf is None:
        return None
    scale, shift, alpha = in_tf
    for t in (scale, shift):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("in_tf: contiguous float32 scale / shift expected")
    return C.byref(InAffine(_ptr(scale), _ptr(shift), _ptr(alpha)))


def conv3d_fwd_kernel_name(x, y, ksize, stride) -> str:
    """kernel family ``conv3d_fwd`` runs for this layer (for benchmark / report labels)"""
    ax, ay = act(x), act(y)
    return lib.segmi_conv3d_fwd_kernel_name(dtype_code(x), C.byref(ax), C.byref(ay), ksize, stride).decode()


def conv3d_in_affine_ok(x, y, ksize, stride) -> bool:
    ax, ay = act(x), act(y)
    return bool(lib.segmi_conv3d_in_affine_ok(dtype_code(x), C.byref(ax), C.byref(ay), ksize, stride))


def conv3d_bn_bwd_sums_ok(x, y, ksize, stride) -> bool:
    ax, ay = act(x), act(y)
    return bool(lib.segmi_conv3d_bn_bwd_sums_ok(dtype_code(x), C.byref(ax), C.byref(ay), ksize, stride))


def wgrad_cus(cus: int) -> int:
    """compute units a weight-gradient call with the argument ``cus`` sizes its grid for (segmi_wgrad_cus:
    a multiple of 8 in [8, 256]; <= 0 = the whole chip; SEGMI_WGRAD_CUS overrides)"""
    return int(lib.segmi_wgrad_cus(int(cus)))


def cu_masked_stream(cus_enabled: int, device=None) -> "torch.cuda.Stream":
    """A torch stream over a HIP stream restricted to the first ``cus_enabled / 8`` CUs of every XCD
    (segmi_stream_create_cumask).  The HIP stream lives as long as the process."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    h = C.c_void_p()
    with torch.cuda.device(dev):
        check(lib.segmi_stream_create_cumask(int(cus_enabled), C.byref(h)), "stream_create_cumask")
    return torch.cuda.ExternalStream(h.value, device=dev)


def _bn_fin(fin):
    """fin = (count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift)
    -> byref(segmi_bn_fin) or None: the launch that writes the statistics rows also finalises them"""
    if fin is None:
        return None
    count, gamma, beta, rm, rv, momentum, eps, mean, invstd, scale, shift = fin
    return C.byref(BnFin(float(count), _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv), float(momentum), float(eps),
                         _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift)))


def _bn_bwd_fin(fin):
    """fin = (count, dgamma, dbeta, dalpha, coef) -> segmi_bn_bwd_fin or None"""
    if fin is None:
        return None
    count, dgamma, dbeta, dalpha, coef = fin
    return BnBwdFin(float(count), _ptr(dgamma), _ptr(dbeta), _ptr(dalpha), _ptr(coef))


def conv3d_fwd(x, y, packed, w_src, w_kind, bias, ksize, stride, prelu_alpha=None,
               residual=None, stats=None, in_tf=None, bn_bwd=None, stats_fin=None, bn_bwd_fin=None) -> None:
    """``in_tf`` = (scale, shift, alpha): the producer's BatchNorm-apply + PReLU is applied to ``x``
    while it is staged (segmi_in_affine; only where ``conv3d_in_affine_ok``).
    ``bn_bwd`` = (x_raw, mean, invstd, gamma, beta, alpha | None, partials): this launch is an
    input-gradient conv whose output flows into that BatchNorm + PReLU; its epilogue also writes the
    partial rows of the BatchNorm-backward reduction (segmi_bn_bwd_sums; where ``conv3d_bn_bwd_sums_ok``).
    ``stats_fin`` / ``bn_bwd_fin``: the launch also finalises the ``stats`` / ``bn_bwd`` rows (``_bn_fin``,
    ``_bn_bwd_fin``): no separate bn_finalize / bn_act_bwd_finalize launch."""
    ax, ay = act(x), act(y)
    ar = act(residual) if residual is not None else None
    bb = None
    if bn_bwd is not None:
        xr, mean, invstd, gamma, beta, alpha, part = bn_bwd
        abx = act(xr)
        bf = _bn_bwd_fin(bn_bwd_fin)
        bb = C.byref(BnBwdSums(C.pointer(abx), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(alpha),
                               _ptr(part), C.pointer(bf) if bf is not None else None))
    check(lib.segmi_conv3d_fwd(dtype_code(x), C.byref(ax), C.byref(ay), _ptr(packed),
                               _ptr(w_src), w_kind, _ptr(bias), _ptr(prelu_alpha), _ref(ar),
                               _ptr(stats), ksize, stride, _in_affine(in_tf), bb, _bn_fin(stats_fin), _stream()),
          "conv3d_fwd")


def dectop_ok(x, y) -> bool:
    ax, ay = act(x), act(y)
    return bool(lib.segmi_dectop_ok(dtype_code(x), C.byref(ax), C.byref(ay)))


def dectop_up_frag(w_t: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """A-operand fragments of the transposed conv for ``segmi_dectop_fwd``: w_t [32, 16, 3, 3, 3] f32
    (torch ConvTranspose3d layout), scale f32[16] (folded BatchNorm) -> bf16 [27, 64, 8]."""
    ws = (w_t * scale.view(1, -1, 1, 1, 1)).reshape(32, 16, 27)
    return ws.permute(2, 0, 1).reshape(27, 4, 8, 16).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16).reshape(27, 64, 8)


def dectop_fwd(x, y, up_frag, up_bias, up_alpha, conv_packed, conv_bias, alpha_in_unit_range=False) -> None:
    """inference: ConvTranspose3d(32 -> 16) + folded BN + PReLU, then conv(16 -> 16) + identity residual,
    one launch (segmi_dectop_fwd)"""
    ax, ay = act(x), act(y)
    check(lib.segmi_dectop_fwd(dtype_code(x), C.byref(ax), C.byref(ay), _ptr(up_frag), _ptr(up_bias),
                               _ptr(up_alpha), int(bool(alpha_in_unit_range)), _ptr(conv_packed), _ptr(conv_bias),
                               _stream()), "dectop_fwd")


def conv3d_pair_ok(x, y_a, y_b) -> bool:
    ax, aa, ab = act(x), act(y_a), act(y_b)
    return bool(lib.segmi_conv3d_pair_ok(dtype_code(x), C.byref(ax), C.byref(aa), C.byref(ab)))


def conv3d_fwd_pair(x, y_a, w_a, bias_a, y_b, w_b, bias_b, stride, prelu_alpha_a=None,
                    stats_a=None, stats_fin_a=None) -> None:
    """Subunit-0 and residual convolution of a small-Cin ResidualUnit in one launch."""
    ax, aa, ab = act(x), act(y_a), act(y_b)
    win = C.byref(x.windows()) if isinstance(x, WindowBatch) else None
    check(lib.segmi_conv3d_fwd_pair(dtype_code(x), C.byref(ax), C.byref(aa), _ptr(w_a), _ptr(bias_a),
                                    _ptr(prelu_alpha_a), _ptr(stats_a), C.byref(ab), _ptr(w_b),
                                    _ptr(bias_b), stride, _bn_fin(stats_fin_a), win, _stream()), "conv3d_fwd_pair")


def conv3d_split_act_ok(x, y, ksize, stride) -> bool:
    ax, ay = act(x), act(y)
    return bool(lib.segmi_conv3d_split_act_ok(dtype_code(x), C.byref(ax), C.byref(ay), ksize, stride))


def conv3d_fwd_split_act(x, y, packed, bias, prelu_alpha, act_channels, ksize, stride, bias_b=None,
                         stats=None, stats_fin=None) -> None:
    """one conv with two weight sets (concatenated pack): PReLU on the first ``act_channels`` only.
    Training: ``prelu_alpha`` None, ``bias_b`` = the second convolution's bias, ``stats`` (+ ``stats_fin``) =
    BatchNorm statistics rows [conv3d_stats_rows][2][act_channels] of the first ``act_channels`` outputs."""
    ax, ay = act(x), act(y)
    check(lib.segmi_conv3d_fwd_split_act(dtype_code(x), C.byref(ax), C.byref(ay), _ptr(packed), _ptr(bias),
                                         _ptr(prelu_alpha), act_channels, ksize, stride, _ptr(bias_b),
                                         _ptr(stats), _bn_fin(stats_fin), _stream()),
          "conv3d_fwd_split_act")


def convT3d_stats_rows(x, y) -> int:
    ax, ay = act(x), act(y)
    return int(lib.segmi_convT3d_stats_rows(dtype_code(x), C.byref(ax), C.byref(ay)))


def convT3d_fwd(x, y, packed, w_src, bias, prelu_alpha=None, residual=None, stats=None,
                stats_fin=None) -> None:
    ax, ay = act(x), act(y)
    ar = act(residual) if residual is not None else None
    check(lib.segmi_convT3d_fwd(dtype_code(x), C.byref(ax), C.byref(ay), _ptr(packed),
                                _ptr(w_src), _ptr(bias), _ptr(prelu_alpha), _ref(ar),
                                _ptr(stats), _bn_fin(stats_fin), _stream()), "convT3d_fwd")


def conv3d_wgrad_workspace(x, dy, ksize, stride, cus: int = 0) -> int:
    """``cus``: the compute-unit budget the call will be made with (it sizes the partial slabs)"""
    ax, ay = act(x), act(dy)
    return int(lib.segmi_conv3d_wgrad_workspace(dtype_code(x), C.byref(ax), C.byref(ay), ksize,
                                                stride, int(cus)))


def conv3d_wgrad(x, dy, dw, db, ksize, stride, workspace, in_tf=None, cus: int = 0) -> None:
    """``cus``: compute units the kernel sizes its grid for (0 = the whole chip); per call, no global state"""
    ax, ay = act(x), act(dy)
    check(lib.segmi_conv3d_wgrad(dtype_code(x), C.byref(ax), C.byref(ay), _ptr(dw), _ptr(db),
                                 ksize, stride, _ptr(workspace), _in_affine(in_tf), int(cus), _stream()),
          "conv3d_wgrad")


def bias_grad(dy, db, workspace) -> None:
    ay = act(dy)
    check(lib.segmi_bias_grad(dtype_code(dy), C.byref(ay), _ptr(db), _ptr(workspace),
                              _stream()), "bias_grad")


# ------------------------------------------------------------------ norm + activation
def bn_stats_rows(x) -> int:
    ax = act(x)
    return int(lib.segmi_bn_stats_rows(C.byref(ax)))


def bn_stats(x, partials) -> None:
    ax = act(x)
    check(lib.segmi_bn_stats(dtype_code(x), C.byref(ax), _ptr(partials), _stream()), "bn_stats")


def bn_finalize(partials, rows, c, count, gamma, beta, running_mean, running_var, momentum,
                eps, mean, invstd, scale, shift) -> None:
    check(lib.segmi_bn_finalize(_ptr(partials), rows, c, float(count), _ptr(gamma), _ptr(beta),
                                _ptr(running_mean), _ptr(running_var), momentum, eps,
                                _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift), _stream()),
          "bn_finalize")


def bn_eval_affine(gamma, beta, running_mean, running_var, eps, scale, shift) -> None:
    check(lib.segmi_bn_eval_affine(running_mean.numel(), _ptr(gamma), _ptr(beta),
                                   _ptr(running_mean), _ptr(running_var), eps, _ptr(scale),
                                   _ptr(shift), _stream()), "bn_eval_affine")


def bn_act_fwd(x, y, scale, shift, prelu_alpha=None, residual=None, dropout=(0.0, 0)) -> None:
    """dropout = (p, seed): ADN dropout between norm and activation (mask recomputed in backward)."""
    ax, ay = act(x), act(y)
    ar = act(residual) if residual is not None else None
    check(lib.segmi_bn_act_fwd(dtype_code(x), C.byref(ax), C.byref(ay), _ptr(scale),
                               _ptr(shift), _ptr(prelu_alpha), _ref(ar), float(dropout[0]),
                               int(dropout[1]) & 0xFFFFFFFF, _stream()),
          "bn_act_fwd")


def bn_act_bwd_rows(x) -> int:
    ax = act(x)
    return int(lib.segmi_bn_act_bwd_rows(C.byref(ax)))


def bn_act_bwd_reduce(dy, x, mean, invstd, gamma, beta, prelu_alpha, partials,
                      dropout=(0.0, 0), fin=None) -> None:
    """``fin`` = (count, dgamma, dbeta, dalpha, coef): finalise in the same launch"""
    ady, ax = act(dy), act(x)
    bf = _bn_bwd_fin(fin)
    check(lib.segmi_bn_act_bwd_reduce(dtype_code(x), C.byref(ady), C.byref(ax), _ptr(mean),
                                      _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(prelu_alpha),
                                      _ptr(partials), float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF,
                                      C.byref(bf) if bf is not None else None, _stream()), "bn_act_bwd_reduce")


def bn_act_bwd_fused_ok(dy, x, dx) -> bool:
    ady, ax, adx = act(dy), act(x), act(dx)
    return bool(lib.segmi_bn_act_bwd_fused_ok(dtype_code(x), C.byref(ady), C.byref(ax), C.byref(adx)))


def bn_act_bwd_fused_rows(x) -> int:
    a = act(x)
    return int(lib.segmi_bn_act_bwd_fused_rows(C.byref(a)))


def bn_act_bwd_fused(dy, x, dx, mean, invstd, gamma, beta, prelu_alpha, partials, fin, max_wgs: int = 0) -> None:
    """reduce + finalise + apply of the BatchNorm / PReLU backward in one launch (small tensors);
    ``fin`` = (count, dgamma, dbeta, dalpha, coef); ``max_wgs``: the most workgroups (= whole CUs) the launch
    may hold while its hand-off completes (0 = what the device holds at once)"""
    ady, ax, adx = act(dy), act(x), act(dx)
    bf = _bn_bwd_fin(fin)
    check(lib.segmi_bn_act_bwd_fused(dtype_code(x), C.byref(ady), C.byref(ax), C.byref(adx), _ptr(mean),
                                     _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(prelu_alpha), _ptr(partials),
                                     C.byref(bf), int(max_wgs), _stream()), "bn_act_bwd_fused")


def bn_act_bwd_fused_wgs(x, max_wgs: int = 0) -> int:
    """workgroups ``bn_act_bwd_fused`` would launch for ``x`` under ``max_wgs`` on the current device"""
    a = act(x)
    return int(lib.segmi_bn_act_bwd_fused_wgs(dtype_code(x), C.byref(a), int(max_wgs)))


def fused_timeouts(reset: bool = False) -> int:
    """expiries of ``bn_act_bwd_fused``'s bounded hand-off wait so far (host-visible counter, no device sync).
    Non-zero: some launch wrote NaN gradients instead of hanging -- see ``check_fused_timeouts``."""
    return int(lib.segmi_fused_timeouts(1 if reset else 0))


def check_fused_timeouts(where: str = "") -> None:
    """raise (and clear the counter) when a one-launch BatchNorm backward gave up waiting: its gradients are NaN, and
    every optimiser step since has poisoned the weights -- the reference stops on a non-finite loss
    (monai_unet.py:512-518); here the failure is an exception instead of a silent NaN run"""
    n = fused_timeouts(reset=True)
    if n:
        raise RuntimeError(
            f"{n} workgroup(s) of segmi_bn_act_bwd_fused gave up waiting for their launch's last workgroup"
            f"{' (' + where + ')' if where else ''}: the launch was never fully resident (GPU shared with another "
            "process, or CUs masked).  The gradients of that step are NaN and the weights are poisoned: restart "
            "from the last checkpoint with SEGMI_FUSE_BN_BWD_SMALL=0 (three launches, no grid-wide wait).")


def fused_test_hook(poll_limit: int = 0, no_publish: bool = False) -> None:
    """tests only: poll bound of the hand-off wait (0 = default) / withhold the flag so every waiter expires"""
    check(lib.segmi_fused_test_hook(int(poll_limit), 1 if no_publish else 0), "fused_test_hook")


def bn_act_bwd_apply_conv_ok(dy, x, dx, out) -> bool:
    ady, ax, adx, ao = act(dy), act(x), act(dx), act(out)
    return bool(lib.segmi_bn_act_bwd_apply_conv_ok(dtype_code(x), C.byref(ady), C.byref(ax), C.byref(adx),
                                                   C.byref(ao)))


def bn_act_bwd_apply_conv(dy, x, dx, mean, invstd, gamma, beta, prelu_alpha, coef, out, packed) -> None:
    """dx = bn_act_bwd_apply(dy, x) and out = conv_k3s2(dx) as ONE launch (the input gradient of a decoder
    level's transposed convolution; csrc/conv_bnbwd_impl.h): same bits as the two calls"""
    ady, ax, adx, ao = act(dy), act(x), act(dx), act(out)
    check(lib.segmi_bn_act_bwd_apply_conv(dtype_code(x), C.byref(ady), C.byref(ax), C.byref(adx), _ptr(mean),
                                          _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(prelu_alpha), _ptr(coef),
                                          C.byref(ao), _ptr(packed), _stream()), "bn_act_bwd_apply_conv")


def bn_act_bwd_finalize(partials, rows, c, count, gamma, invstd, dgamma, dbeta, dalpha,
                        coef) -> None:
    check(lib.segmi_bn_act_bwd_finalize(_ptr(partials), rows, c, float(count), _ptr(gamma),
                                        _ptr(invstd), _ptr(dgamma), _ptr(dbeta), _ptr(dalpha),
                                        _ptr(coef), _stream()), "bn_act_bwd_finalize")


def bn_act_bwd_apply(dy, x, dx, mean, invstd, gamma, beta, prelu_alpha, coef,
                     dropout=(0.0, 0)) -> None:
    ady, ax, adx = act(dy), act(x), act(dx)
    check(lib.segmi_bn_act_bwd_apply(dtype_code(x), C.byref(ady), C.byref(ax), C.byref(adx),
                                     _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta),
                                     _ptr(prelu_alpha), _ptr(coef), float(dropout[0]),
                                     int(dropout[1]) & 0xFFFFFFFF, _stream()),
          "bn_act_bwd_apply")


def add(a, b, out) -> None:
    aa, ao = act(a), act(out)
    ab = act(b) if b is not None else None
    check(lib.segmi_add(dtype_code(a), C.byref(aa), _ref(ab), C.byref(ao), _stream()), "add")


def cast_copy(src, dst) -> None:
    a, b = act(src), act(dst)
    check(lib.segmi_cast_copy(dtype_code(src), C.byref(a), dtype_code(dst), C.byref(b),
                              _stream()), "cast_copy")


def nchw_to_ndhwc(src: torch.Tensor, dst: torch.Tensor) -> None:
    """src f32 [N,C,D,H,W] contiguous -> dst NDHWC (f32 or bf16)."""
    if src.dtype != torch.float32 or not src.is_contiguous():
        raise ValueError("nchw_to_ndhwc expects a contiguous float32 NCDHW tensor")
    b = act(dst)
    check(lib.segmi_nchw_to_ndhwc(_ptr(src), dtype_code(dst), C.byref(b), _stream()),
          "nchw_to_ndhwc")


def ndhwc_to_nchw(src: torch.Tensor, dst: torch.Tensor) -> None:
    a = act(src)
    if dst.dtype != torch.float32 or not dst.is_contiguous():
        raise ValueError("ndhwc_to_nchw writes a contiguous float32 NCDHW tensor")
    check(lib.segmi_ndhwc_to_nchw(dtype_code(src), C.byref(a), _ptr(dst), _stream()),
          "ndhwc_to_nchw")


# ------------------------------------------------------------------ loss + optimiser
def dice_chunks(logits) -> int:
    a = act(logits)
    return int(lib.segmi_dice_chunks(C.byref(a)))


def softmax_dice_fwd(logits, labels, partials, coef, loss, smooth_nr=1e-5, smooth_dr=1e-5):
    a = act(logits)
    check(lib.segmi_softmax_dice_fwd(dtype_code(logits), C.byref(a), _ptr(labels),
                                     _ptr(partials), _ptr(coef), _ptr(loss), smooth_nr,
                                     smooth_dr, _stream()), "softmax_dice_fwd")


def softmax_dice_bwd(logits, labels, coef, grad_scale, dlogits, scratch=None, bias_grad=None) -> None:
    """bias_grad (f32[K], optional): channel sums of dlogits, folded into the same pass; `scratch`
    is then the forward's partials buffer."""
    a, b = act(logits), act(dlogits)
    check(lib.segmi_softmax_dice_bwd(dtype_code(logits), C.byref(a), _ptr(labels), _ptr(coef),
                                     float(grad_scale), C.byref(b), _ptr(scratch), _ptr(bias_grad),
                                     _stream()),
          "softmax_dice_bwd")


def adam_step(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, lr, beta1, beta2, eps,
              weight_decay, step, grad_scale=1.0) -> None:
    check(lib.segmi_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                              _ptr(max_exp_avg_sq), param.numel(), lr, beta1, beta2, eps,
                              weight_decay, step, grad_scale, _stream()), "adam_step")


def sgd_step(param, grad, buf, lr, momentum, weight_decay, first_step, grad_scale=1.0) -> None:
    check(lib.segmi_sgd_step(_ptr(param), _ptr(grad), _ptr(buf), param.numel(), lr, momentum,
                             weight_decay, int(first_step), grad_scale, _stream()), "sgd_step")


def adabelief_step(param, grad, exp_avg, exp_avg_var, lr, beta1, beta2, eps, weight_decay,
                   weight_decouple, step, grad_scale=1.0) -> None:
    check(lib.segmi_adabelief_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_var),
                                   param.numel(), lr, beta1, beta2, eps, weight_decay,
                                   int(weight_decouple), step, grad_scale, _stream()),
          "adabelief_step")


# ------------------------------------------------------------------ sliding window
def _starts(starts: Sequence[Sequence[int]], width: int):
    arr = np.ascontiguousarray(np.asarray(starts, dtype=np.int32).reshape(-1, width))
    return arr, arr.ctypes.data_as(C.c_void_p)


SW_MAX_VIEWS = 32       # window views per first-layer call (segmi_windows.offset in include/segmi.h)
SW_MAX_WINDOWS = 16     # windows per segmi_sw_gather / segmi_sw_scatter_add call (kMaxWin in sliding.hip)


def sw_gather(image, img_index, starts, windows) -> None:
    a = act(image)
    for i in range(0, len(starts), SW_MAX_WINDOWS):      # larger groups: one launch per 16 windows
        arr, p = _starts(starts[i:i + SW_MAX_WINDOWS], 3)
        b = act(windows[i:i + arr.shape[0]])
        check(lib.segmi_sw_gather(dtype_code(image), C.byref(a), img_index, p, arr.shape[0],
                                  dtype_code(windows), C.byref(b), _stream()), "sw_gather")


def sw_scatter_add(pred, starts, acc, cnt, importance=None) -> None:
    b = act(acc)
    # larger groups: one launch per 16 windows, in schedule order (launches on one stream run in
    # order, so every voxel still receives its windows in the reference's sequence)
    for i in range(0, len(starts), SW_MAX_WINDOWS):
        arr, p = _starts(starts[i:i + SW_MAX_WINDOWS], 3)
        a = act(pred[i:i + arr.shape[0]])
        check(lib.segmi_sw_scatter_add(dtype_code(pred), C.byref(a), p, arr.shape[0],
                                       _ptr(importance), C.byref(b), _ptr(cnt), _stream()),
              "sw_scatter_add")


_LABEL_BYTES = {torch.uint8: 1, torch.int16: 2, torch.int32: 4}


def sw_finalize(acc, cnt, labels, write_logits=True) -> None:
    a = act(acc)
    check(lib.segmi_sw_finalize(C.byref(a), _ptr(cnt), int(write_logits), _ptr(labels),
                                _LABEL_BYTES[labels.dtype], _stream()), "sw_finalize")


def sw_blend(cache, starts_zyx, win_lo, win_hi, roi, d, h, w, importance=None, out_logits=None,
             out_count=None, labels=None, normalize=True) -> None:
    """cache [slots, rd, rh, rw, K] (NDHWC, bf16/f32); starts_zyx = three ascending origin lists."""
    _require_device(cache)
    if cache.dim() != 5 or cache.stride(4) != 1 or not cache.is_contiguous():
        raise ValueError("sw_blend: cache must be a contiguous [slots, rd, rh, rw, K] tensor")
    arrs = [np.ascontiguousarray(np.asarray(sv, dtype=np.int32)) for sv in starts_zyx]
    k = cache.shape[4]
    ldo = out_logits.stride(-2) if out_logits is not None else 0
    if out_logits is not None and (out_logits.dtype != torch.float32 or out_logits.stride(-1) != 1):
        raise ValueError("sw_blend: out_logits must be float32 NDHWC")
    check(lib.segmi_sw_blend(
        dtype_code(cache), _ptr(cache), k, cache.stride(3),
        arrs[0].ctypes.data_as(C.c_void_p), len(arrs[0]), arrs[1].ctypes.data_as(C.c_void_p),
        len(arrs[1]), arrs[2].ctypes.data_as(C.c_void_p), len(arrs[2]), int(win_lo), int(win_hi),
        int(roi[0]), int(roi[1]), int(roi[2]), _ptr(importance), int(d), int(h), int(w),
        _ptr(out_logits), int(ldo), _ptr(out_count), _ptr(labels),
        _LABEL_BYTES[labels.dtype] if labels is not None else 1, int(bool(normalize)), _stream()),
        "sw_blend")


def argmax(logits, labels) -> None:
    a = act(logits)
    check(lib.segmi_argmax(dtype_code(logits), C.byref(a), _ptr(labels),
                           _LABEL_BYTES[labels.dtype], _stream()), "argmax")


def label_counts(pred, truth, k, counts) -> None:
    check(lib.segmi_label_counts(_ptr(pred), _ptr(truth), pred.numel(), k, _ptr(counts),
                                 _stream()), "label_counts")


# ------------------------------------------------------------------ image ops
_PIXEL = {torch.float32: 0, torch.uint8: 1, torch.int16: 2, torch.int32: 3, torch.uint16: 4}


def resample3d(src: torch.Tensor, out_size_zyx, index_map, nearest=False, default=0.0, border=False,
               half_even=False):
    """src [z,y,x] -> dst [z,y,x]; index_map: 3x4 out-index(x,y,z,1) -> in-index(x,y,z).
    ``border``: clamp the continuous index to the buffer (MONAI ``padding_mode="border"``) instead
    of ITK's default-pixel-outside rule.  ``half_even`` (nearest only): round x.5 to the even index
    (torch ``grid_sample`` / MONAI) instead of up (ITK)."""
    _require_device(src)
    if src.dim() != 3 or not src.is_contiguous():
        raise ValueError("resample3d expects a contiguous [z,y,x] tensor")
    dz, dy, dx = (int(v) for v in out_size_zyx)
    dst = torch.empty((dz, dy, dx), dtype=src.dtype, device=src.device)
    m = np.ascontiguousarray(np.asarray(index_map, dtype=np.float64).reshape(12))
    sz, sy, sx = src.shape
    check(lib.segmi_resample3d(_PIXEL[src.dtype], _ptr(src), sx, sy, sz, _ptr(dst), dx, dy, dz,
                               m.ctypes.data_as(C.c_void_p), (1 if nearest else 0) | (2 if border else 0) | (4 if nearest and half_even else 0),
                               float(default), _stream()), "resample3d")
    return dst


def normalize_intensity_(x: torch.Tensor) -> torch.Tensor:
    """in-place channel-wise (x-mean)/std of a contiguous f32 [C, ...] tensor."""
    _require_device(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("normalize_intensity_ expects contiguous float32")
    c = x.shape[0]
    nvox = x.numel() // c
    ws = torch.empty(int(lib.segmi_normalize_workspace(c, nvox)), dtype=torch.uint8,
                     device=x.device)
    check(lib.segmi_normalize_intensity(_ptr(x), c, nvox, _ptr(ws), _stream()),
          "normalize_intensity")
    return x


def crop_patches(image, label, starts, flips, out_image, out_label) -> None:
    a, b = act(image), act(out_image)
    arr, p = _starts(starts, 4)
    fl = None
    flp = None
    if flips is not None:
        fl = np.ascontiguousarray(np.asarray(flips, dtype=np.uint8))
        flp = fl.ctypes.data_as(C.c_void_p)
    check(lib.segmi_crop_patches(C.byref(a), _ptr(label), p, flp, arr.shape[0],
                                 dtype_code(out_image), C.byref(b), _ptr(out_label),
                                 _stream()), "crop_patches")


def warp_crop_patches(image, label, starts, flips, index_map, out_image, out_label) -> None:
    """crop_patches with a 3x4 affine (augmented index (x,y,z,1) -> source index) composed in."""
    a, b = act(image), act(out_image)
    arr, p = _starts(starts, 4)
    flp = None
    if flips is not None:
        fl = np.ascontiguousarray(np.asarray(flips, dtype=np.uint8))
        flp = fl.ctypes.data_as(C.c_void_p)
    m = np.ascontiguousarray(np.asarray(index_map, dtype=np.float64).reshape(12))
    check(lib.segmi_warp_crop_patches(C.byref(a), _ptr(label), p, flp, arr.shape[0],
                                      m.ctypes.data_as(C.c_void_p), dtype_code(out_image),
                                      C.byref(b), _ptr(out_label), _stream()), "warp_crop_patches")


def intensity_augment(patches, contrast=None, hist=None, bias=None) -> None:
    """In-place RandAdjustContrast / RandHistogramShift / RandBiasField on f32 NDHWC patches.

    contrast = (on uint8[n], gamma f32[n]); hist = (on, ctrl f32[n][k]); bias = (on, coef f32[n][20])."""
    _require_device(patches)
    if patches.dtype != torch.float32 or patches.dim() != 5 or not patches.is_contiguous():
        raise ValueError("intensity_augment: dense float32 [n, d, h, w, c] patches expected")
    n, rd, rh, rw, c = patches.shape
    ws = torch.empty(int(lib.segmi_intensity_workspace(n)), dtype=torch.uint8, device=patches.device)
    keep = []

    def arr(x, dt):
        if x is None:
            return None
        a = np.ascontiguousarray(np.asarray(x, dtype=dt))
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    con, gam = (contrast if contrast is not None else (None, None))
    hon, ctl = (hist if hist is not None else (None, None))
    bon, cof = (bias if bias is not None else (None, None))
    nctrl = int(np.asarray(ctl).shape[1]) if ctl is not None else 0
    check(lib.segmi_intensity_augment(_ptr(patches), n, rd, rh, rw, c, arr(con, np.uint8),
                                      arr(gam, np.float32), arr(hon, np.uint8), arr(ctl, np.float32),
                                      nctrl, arr(bon, np.uint8), arr(cof, np.float32), _ptr(ws),
                                      _stream()), "intensity_augment")


def kspace_augment(patches, gibbs=None, spike=None) -> None:
    """In-place RandGibbsNoise / RandKSpaceSpikeNoise on f32 NDHWC patches.

    gibbs = (on uint8[n], alpha f32[n]); spike = (on uint8[n], loc int32[n][3] (z,y,x), u f32[n])."""
    _require_device(patches)
    if patches.dtype != torch.float32 or patches.dim() != 5 or not patches.is_contiguous():
        raise ValueError("kspace_augment: dense float32 [n, d, h, w, c] patches expected")
    n, rd, rh, rw, c = patches.shape
    gon, alpha = gibbs if gibbs is not None else (None, None)
    son, loc, u = spike if spike is not None else (None, None, None)
    nsel = max(int(np.asarray(gon).astype(bool).sum()) if gon is not None else 0,
               int(np.asarray(son).astype(bool).sum()) if son is not None else 0)
    if nsel == 0:
        return
    ws = torch.empty(int(lib.segmi_kspace_workspace(n, rd, rh, rw)), dtype=torch.uint8,
                     device=patches.device)
    keep = []

    def arr(x, dt):
        if x is None:
            return None
        a = np.ascontiguousarray(np.asarray(x, dtype=dt))
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    check(lib.segmi_kspace_augment(_ptr(patches), n, rd, rh, rw, c, arr(gon, np.uint8),
                                   arr(alpha, np.float32), arr(son, np.uint8), arr(loc, np.int32),
                                   arr(u, np.float32), _ptr(ws), _stream()), "kspace_augment")


def _ptr_table(tensors, dtype):
    for t in tensors:
        _require_device(t)
        if t.dtype != dtype or not t.is_contiguous():
            raise ValueError(f"ensemble: contiguous {dtype} tensors expected")
    tab = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return tab


def ensemble_mean(logits, weights, out) -> None:
    """out = mean_e(logits[e] * w[e] / mean(w)) over same-shaped f32 tensors (MONAI MeanEnsemble)."""
    tab = _ptr_table(list(logits) + [out], torch.float32)
    w = np.ascontiguousarray(np.asarray(weights, dtype=np.float32)) if weights is not None else None
    check(lib.segmi_ensemble_mean(tab, w.ctypes.data_as(C.c_void_p) if w is not None else None,
                                  len(logits), out.numel(), _ptr(out), _stream()), "ensemble_mean")


def ensemble_vote(labels, out) -> None:
    tab = _ptr_table(list(labels) + [out], torch.int32)
    check(lib.segmi_ensemble_vote(tab, len(labels), out.numel(), _ptr(out), _stream()), "ensemble_vote")


def ensemble_select(labels, tissue_model: dict, out) -> None:
    """SelectBestEnsemble: tissue_model = {tissue id: model index}, applied in dict order."""
    tab = _ptr_table(list(labels) + [out], torch.int32)
    ts = np.ascontiguousarray(np.asarray(list(tissue_model.keys()), dtype=np.int32))
    ms = np.ascontiguousarray(np.asarray(list(tissue_model.values()), dtype=np.int32))
    check(lib.segmi_ensemble_select(tab, len(labels), ts.ctypes.data_as(C.c_void_p),
                                    ms.ctypes.data_as(C.c_void_p), len(ts), out.numel(), _ptr(out),
                                    _stream()), "ensemble_select")
