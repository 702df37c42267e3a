"""Sliding-window inference on device.

Replaces ``monai.inferers.sliding_window_inference`` / ``SlidingWindowInferer`` as called at
reference ``src/segmantic/seg/monai_unet.py:354-356`` (validation: roi 160^d, sw_batch 4),
``:637-639,665`` (predict) and ``:840-842`` (ensemble: roi 96, overlap 0.5).

Window schedule (host, integer arithmetic -- identical to MONAI's ``dense_patch_slices``):
  image smaller than roi -> symmetric zero padding (half = diff // 2 low);
  interval = roi if roi == img else max(int(roi * (1 - overlap)), 1);
  n = ceil((img - roi) / interval) + 1 ; start_k = min(k * interval, img - roi);
  windows enumerated with the first spatial dim slowest, groups of ``sw_batch_size``.
Data path (device): gather kernel -> predictor -> ordered blend -> divide (+ fused argmax).  The
blend adds the windows covering a voxel in schedule order in f32, so its result is bit-identical
to the reference's sequential ``out[slice] += w * pred``; it runs either deferred (all window
predictions kept in HBM, one blend pass, ``segmi_sw_blend``) or streaming (f32 accumulator
read-modify-written per window group, ``segmi_sw_scatter_add``).
"""
from __future__ import annotations

import math
import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from .. import ops
from . import streams
from .losses import as_ndhwc


def scan_intervals(image_size, roi, overlap) -> List[int]:
    out = []
    for i, r in zip(image_size, roi):
        if r == i:
            out.append(int(r))
        else:
            iv = int(r * (1 - overlap))
            out.append(iv if iv > 0 else 1)
    return out


def dense_starts(image_size, roi, overlap) -> List[List[int]]:
    """Per-dimension window origins (ascending) of MONAI's dense schedule."""
    iv = scan_intervals(image_size, roi, overlap)
    per_dim = []
    for i, r, s in zip(image_size, roi, iv):
        n = int(math.ceil(float(i - r) / s)) + 1
        per_dim.append([min(k * s, i - r) for k in range(n)])
    return per_dim


def window_starts(image_size, roi, overlap) -> List[Tuple[int, ...]]:
    wins = [()]
    for starts in dense_starts(image_size, roi, overlap):
        wins = [w + (s,) for w in wins for s in starts]
    return wins


_IMPORTANCE: dict = {}


def gaussian_importance(roi, sigma_scale=0.125, device="cpu") -> torch.Tensor:
    """MONAI's gaussian importance map (product of per-axis gaussians, clamped).  Built on the host in MONAI's
    order of operations (the blend must reproduce its bits) and kept per (roi, sigma, device): 2 M exponentials and
    an 8 MB upload per call otherwise -- ~10 ms of host time in front of every volume's first launch."""
    key = (tuple(int(r) for r in roi), float(sigma_scale), str(torch.device(device)))
    hit = _IMPORTANCE.get(key)
    if hit is not None:
        return hit
    if len(_IMPORTANCE) >= 8:
        _IMPORTANCE.clear()
    w = _gaussian_importance_host(roi, sigma_scale).to(device)
    _IMPORTANCE[key] = w
    return w


def _gaussian_importance_host(roi, sigma_scale) -> torch.Tensor:
    w = torch.ones(tuple(roi), dtype=torch.float32)
    for d, r in enumerate(roi):
        sigma = sigma_scale * r
        x = torch.arange(-(r - 1) / 2.0, (r - 1) / 2.0 + 1, dtype=torch.float32)
        g = torch.exp(x ** 2 / (-2 * sigma ** 2))
        shp = [1] * len(roi)
        shp[d] = r
        w = w * g.reshape(shp)
    mn = max(float(w[w != 0].min()), 1e-3)
    return torch.clamp(w, min=mn).contiguous()


class SlidingWindowResult:
    """logits: logical [B,K,D,H,W] f32 view of the NDHWC accumulator; labels [B,1,D,H,W]."""

    def __init__(self, logits, labels, count):
        self.logits, self.labels, self.count = logits, labels, count




def default_lanes() -> int:
    """Streams that window groups alternate over when the predictor is this build's own network
    (SEGMI_SW_LANES; 1 = everything on the caller's stream).

    Default 1 since round 3.  Two lanes are worth +0 .. +4 % on a warm box (21.4 vs 20.5, 21.3 vs 21.2,
    22.9 vs 22.6 volumes/s in same-process A/Bs) but, in the first minute of a box's life, the two-lane
    schedule repeatedly ran host-bound at 5 volumes/s (168 ms instead of 8 ms to enqueue a volume's window
    groups, lane busy times unchanged, the one-lane run seconds later in the same process at 20+) -- the
    round-2 driver figure of 17.5 against the builder's 21.3.  One lane does not have that failure mode;
    bench.py reports both.  (Later in round 3 the slow first volumes were traced to allocations of the prediction
    cache, not to the lane count, and removed: `_cache_workspace`.  Two lanes now measure +1.5 .. +3 % whichever
    count is timed first, three lanes +5 % (`gpurun_out/r3/lanes_order.txt`, `lanes_n.txt`).  The default stays 1:
    with several lanes a launch's duration is that of two co-running launches, which makes per-kernel rooflines
    of the line unreadable; `lanes=` / SEGMI_SW_LANES for throughput.)

    Round 4: default 3.  With the forwards as they are now (GPU busy 41.3 of 41.9 ms per volume on one lane, but
    the 16-window launches of the deep levels and the ramps of the 512-workgroup launches leave CUs idle inside
    that time) three lanes take 39.6 ms per volume against 41.1 (2: 40.3, 4: 40.5; `gpurun_out/r4/sweep_063254.txt`),
    labels bit-identical.  bench.py times the default and ALSO a one-lane pass, and takes the per-kernel roofline of
    its inference object from the one-lane pass (a launch timed beside copies of itself says nothing about the
    kernel); profiling scripts pin SEGMI_SW_LANES=1."""
    if os.environ.get("SEGMI_SERIAL"):
        return 1
    return max(1, int(os.environ.get("SEGMI_SW_LANES", "3")))


def _lane_streams(device, n: Optional[int] = None):
    """``n`` streams for alternating window groups (None: ``default_lanes()``; < 2: no side streams,
    everything stays on the current stream, e.g. for per-kernel profiling)."""
    n = default_lanes() if n is None else int(n)
    if n < 2:
        return None
    # the package's shared side streams (seg/streams.py): slots 0 .. 2, then 4 + (slot 3 is the blend's)
    return [streams.shared_stream(device, i if i < streams.BLEND else i + 1) for i in range(n)]




def _blend_stream(device):
    return streams.shared_stream(device, streams.BLEND)


def group_factor() -> int:
    """sw_batch_size multiplier for window groups handed to this build's own network
    (SEGMI_SW_GROUP, default 4: 18.4 -> 20.8 volumes/s on the 512^3 benchmark; 2 -> 19.9)"""
    return max(1, int(os.environ.get("SEGMI_SW_GROUP", "4")))


# The prediction cache of the deferred blend (every window's logits until they are blended: 23 GB for a 512^3
# volume with 16 classes) is scratch of the driver, never handed to the caller.  It is kept across calls, one per
# device: taken from torch's caching allocator per call, the freed block gets split by the smaller allocations in
# between (count map, labels) and every few volumes a call pays a fresh 23 GB hipMalloc -- 0.5 s against 44 ms for
# the whole volume (bench.py caught it as one timed volume in ten).  SEGMI_SW_KEEP_CACHE=0: allocate per call;
# ``release_workspaces()`` returns the memory -- ``predict()`` calls it when its volumes are done, the fit loop after
# every validation epoch, so the cache is never pinned beside a training step's activations.
_CACHE_WS: dict = {}


def _cache_key(dev):
    """one key for get and set: the device index (an index-less ``cuda`` device means the current one)"""
    dev = torch.device(dev)
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _cache_workspace(dev, shape, dtype):
    """(buffer, event of the last call that used it or None)"""
    dev = torch.device(dev)
    if dev.type != "cuda" or os.environ.get("SEGMI_SW_KEEP_CACHE", "1") == "0":
        return torch.empty(shape, dtype=dtype, device=dev), None
    key = _cache_key(dev)
    hit = _CACHE_WS.get(key)
    if hit is not None and tuple(hit[0].shape) == tuple(shape) and hit[0].dtype == dtype:
        return hit[0], hit[1]
    _CACHE_WS.pop(key, None)          # another shape: one live cache per device
    buf = torch.empty(shape, dtype=dtype, device=dev)
    _CACHE_WS[key] = [buf, None]
    return buf, None


def _cache_release_point(dev, stream):
    """the call is done with the cache once `stream` gets here: the next call (possibly on another stream) waits"""
    dev = torch.device(dev)
    hit = _CACHE_WS.get(_cache_key(dev)) if dev.type == "cuda" else None
    if hit is not None:
        ev = torch.cuda.Event()
        ev.record(stream)
        hit[1] = ev


def release_workspaces() -> None:
    """give the kept prediction cache(s) back to the allocator"""
    _CACHE_WS.clear()


def _cache_budget_bytes(device) -> int:
    free, _total = torch.cuda.mem_get_info(device)
    return int(free * 0.6)


def sliding_window_inference(inputs: torch.Tensor, roi_size: Sequence[int], sw_batch_size: int,
                             predictor: Callable[[torch.Tensor], torch.Tensor],
                             overlap: float = 0.25, mode: str = "constant",
                             sigma_scale: float = 0.125, device=None,
                             return_labels: bool = False, window_dtype: Optional[torch.dtype] = None,
                             window_range: Optional[Tuple[int, int]] = None,
                             blend: str = "auto", return_logits: bool = True,
                             z_slab: Optional[Tuple[int, int]] = None, lanes: Optional[int] = None,
                             stats: Optional[dict] = None):
    """inputs [B,C,D,H,W] float32 on the GPU.  ``predictor`` maps [b,C,*roi] -> [b,K,*roi].

    Returns logits [B,K,D,H,W] (float32), or a ``SlidingWindowResult`` when ``return_labels``
    (``return_logits=False`` then skips the f32 logits volume).  ``window_range`` restricts the
    schedule to windows [lo, hi) (multi-GPU sharding; the result is then the un-normalised sum).

    ``blend``: "deferred" keeps every window prediction in HBM (5.4x the logits volume at overlap
    0.5, in the predictor's dtype) and blends once -- each prediction is read exactly once;
    "stream" read-modify-writes an f32 accumulator per window group (the reference's data flow);
    "auto" picks deferred when the cache fits in 60 % of the free HBM.  Both add the windows of a
    voxel in schedule order in f32, so their results are bit-identical.

    ``z_slab`` = (z0, z1): compute only planes [z0, z1) of the first spatial dimension -- the
    windows that intersect the slab are run (all of them, so every voxel of the slab gets its
    complete ordered sum: the slab is bit-identical to the same planes of the full result).  This
    is the multi-GPU form of one volume: rank r takes slab r of ``z_slabs`` and only the label
    slabs travel (``gather_label_slabs``); needs the deferred blend.

    ``lanes``: number of streams the window groups alternate over (own network only; None =
    ``default_lanes()``).  ``stats``: a dict that receives measurement hooks of this call --
    ``host_enqueue_s`` (host time to enqueue all window groups), ``lane_events`` (per lane, one
    (start, end) HIP-event pair per window group), ``lanes``.
    """
    if inputs.dim() == 4:          # 2-D images: run as depth-1 volumes through the same kernels
        own = hasattr(getattr(predictor, "__self__", predictor), "forward_into")
        pred3 = predictor if own else (lambda w: predictor(w.squeeze(2)).unsqueeze(2))
        res = sliding_window_inference(
            inputs.unsqueeze(2), (1,) + tuple(roi_size), sw_batch_size, pred3, overlap, mode,
            sigma_scale, device, return_labels, window_dtype, window_range, blend, return_logits, None,
            lanes, stats)
        if not return_labels:
            return res.squeeze(2)
        return SlidingWindowResult(None if res.logits is None else res.logits.squeeze(2),
                                   None if res.labels is None else res.labels.squeeze(2),
                                   res.count.squeeze(1))
    if inputs.dim() != 5:
        raise ValueError("sliding_window_inference expects [B,C,D,H,W] (or [B,C,H,W])")
    if not inputs.is_cuda:
        raise RuntimeError("segmantic_amd sliding-window inference runs on the GPU only")
    if not 0 <= overlap < 1:
        raise ValueError("overlap must be >= 0 and < 1.")
    if blend not in ("auto", "deferred", "stream"):
        raise ValueError(f"unsupported blend strategy {blend}")
    B, Cin = inputs.shape[0], inputs.shape[1]
    orig = list(inputs.shape[2:])
    roi = [int(r) if r else int(o) for r, o in zip(roi_size, orig)]
    image_size = [max(o, r) for o, r in zip(orig, roi)]
    pad_lo = [max(r - o, 0) // 2 for o, r in zip(orig, roi)]
    per_dim = dense_starts(image_size, roi, overlap)
    wins = [()]
    for starts in per_dim:
        wins = [w + (s,) for w in wins for s in starts]
    lo, hi = (0, len(wins)) if window_range is None else window_range
    out_d, z_off = orig[0], 0
    if z_slab is not None:
        if window_range is not None or blend == "stream":
            raise ValueError("z_slab needs the deferred blend and excludes window_range")
        z_off, z1 = int(z_slab[0]), int(z_slab[1])
        if not 0 <= z_off < z1 <= orig[0]:
            raise ValueError(f"z_slab {z_slab} outside the volume depth {orig[0]}")
        ks = [k for k, s0 in enumerate(per_dim[0]) if s0 - pad_lo[0] < z1 and s0 - pad_lo[0] + roi[0] > z_off]
        nyx = len(per_dim[1]) * len(per_dim[2])
        lo, hi = ks[0] * nyx, (ks[-1] + 1) * nyx
        out_d = z1 - z_off
        blend = "deferred"
    # window origins in un-padded image coordinates (the gather kernel zero-fills outside)
    wins_u = [tuple(s - p for s, p in zip(w, pad_lo)) for w in wins]
    per_dim_u = [[s - p for s in lst] for lst, p in zip(per_dim, pad_lo)]
    per_dim_b = [[s - z_off for s in per_dim_u[0]], per_dim_u[1], per_dim_u[2]]   # blend coordinates
    img = inputs.float()
    img = img.view(B, orig[0], orig[1], orig[2], 1) if Cin == 1 and img.is_contiguous() else as_ndhwc(img)
    imp = None
    if mode == "gaussian":
        imp = gaussian_importance(roi, sigma_scale, inputs.device).view(-1)
    elif mode != "constant":
        raise ValueError(f"unsupported blend mode {mode}")
    dev = inputs.device
    owner = getattr(predictor, "__self__", predictor)
    into = getattr(owner, "forward_into", None)
    if window_dtype is None:    # windows in the network's compute dtype: no separate cast pass
        window_dtype = getattr(owner, "compute_dtype", None) or torch.float32
    partial = window_range is not None
    want_logits = return_logits or not return_labels or partial
    nvox_roi = roi[0] * roi[1] * roi[2]
    outs, labs, cnts = [], [], []
    lanes = _lane_streams(dev, lanes) if into is not None else None
    lane_events = [[] for _ in (lanes or [None])]
    import time as _time
    t_enq = _time.perf_counter()
    # Window groups handed to OUR network may be larger than sw_batch_size: in eval mode every
    # window is computed independently of its batch neighbours (folded BatchNorm: the grouping can
    # only change which kernel a layer picks, i.e. the f32 summation order in front of a bf16
    # rounding), and the deep 8^3 / 16^3 layers of a 4-window forward do not fill the chip (47 us
    # per launch for 3 % of the FLOPs).  A foreign predictor callable always sees exactly
    # sw_batch_size windows, as MONAI would give it.
    # (The streaming fallback keeps the enlarged groups: gather and scatter-add chunk them by 16.)
    if into is not None:
        sw_batch_size = int(sw_batch_size) * group_factor()
    # Windows read in place (own network, one input channel, no padding, 4-element aligned rows and origins):
    # the volume is cast to the compute dtype once and the first-layer kernel takes the windows as views
    # (ops.WindowBatch / segmi_windows) -- no gather pass, no window batch in HBM.  SEGMI_SW_VIEWS=0: gather.
    views_ok = (into is not None and Cin == 1 and not any(pad_lo) and all(o >= r for o, r in zip(orig, roi))
                and os.environ.get("SEGMI_SW_VIEWS", "1") != "0" and sw_batch_size <= ops.SW_MAX_VIEWS
                and ops.WindowBatch.eligible(orig, wins_u[lo:hi], roi)
                and getattr(owner, "window_views_ok", lambda d: False)(window_dtype))
    for b in range(B):
        acc = cnt = cache = None
        K = None
        deferred = False
        main = torch.cuda.current_stream(dev)
        forked = False
        volc = img[b].reshape(orig).to(window_dtype) if views_ok else None
        # Pipelined blend (own network, one lane, whole volume): the windows run z-level by z-level, so once
        # the last window of z-level k is done every output plane below the next level's origin is final --
        # that slab is blended (same kernel, same ordered sums: bit-identical) on a second stream while the
        # forwards of the following levels run.  The blend is HBM-bound, the forwards are issue-bound: of the
        # 6.1 ms the one-shot blend of a 512^3 volume takes, only the last slab's share stays exposed.
        pipe = (into is not None and not partial and z_slab is None
                and len(per_dim[0]) >= 2 and os.environ.get("SEGMI_SW_PIPE_BLEND", "1") != "0")
        nyx = len(per_dim[1]) * len(per_dim[2])
        z_done, next_level = 0, 0
        lab_p = acc_p = cnt_p = None
        def blend_finished_levels(last_window):
            """blend every slab whose covering windows are all enqueued (windows <= last_window)"""
            nonlocal z_done, next_level, lab_p, acc_p, cnt_p
            if not (pipe and deferred and cache is not None):
                return
            nz = len(per_dim_u[0])
            while next_level < nz and last_window >= (next_level + 1) * nyx - 1:
                z1 = per_dim_u[0][next_level + 1] if next_level + 1 < nz else orig[0]
                z1 = max(0, min(int(z1), orig[0]))
                next_level += 1
                if z1 <= z_done:
                    continue
                if cnt_p is None:
                    cnt_p = torch.empty((orig[0], orig[1], orig[2]), dtype=torch.float32, device=dev)
                    if want_logits:
                        acc_p = torch.empty((1, orig[0], orig[1], orig[2], K), dtype=torch.float32, device=dev)
                    if return_labels:
                        lab_p = torch.empty((orig[0], orig[1], orig[2]),
                                            dtype=torch.uint8 if K <= 256 else torch.int32, device=dev)
                bs = _blend_stream(dev)
                # every window of the finished levels has been enqueued (groups go out in schedule order): the blend
                # waits for the current tail of the stream(s) that run them -- the caller's, or every lane's
                for src in (lanes if (lanes is not None and forked) else [main]):
                    ev = torch.cuda.Event()
                    ev.record(src)
                    bs.wait_event(ev)
                with torch.cuda.stream(bs):
                    ops.sw_blend(cache, [[s0 - z_done for s0 in per_dim_u[0]], per_dim_u[1], per_dim_u[2]], lo, hi,
                                 roi, z1 - z_done, orig[1], orig[2], importance=imp,
                                 out_logits=acc_p[:, z_done:z1] if acc_p is not None else None,
                                 out_count=cnt_p[z_done:z1], labels=lab_p[z_done:z1] if lab_p is not None else None,
                                 normalize=True)
                z_done = z1

        def pick_strategy(k, cdtype):
            """deferred blend (every prediction kept until the ordered blend) or the streaming accumulator"""
            nonlocal K, cache, deferred, acc, cnt
            K = int(k)
            need = (hi - lo) * nvox_roi * K * torch.empty((), dtype=cdtype).element_size()
            cshape = (hi - lo, roi[0], roi[1], roi[2], K)
            kept = _CACHE_WS.get(_cache_key(dev)) if dev.type == "cuda" else None
            have = kept is not None and tuple(kept[0].shape) == cshape and kept[0].dtype == cdtype
            deferred = blend != "stream" and max(len(v) for v in per_dim) <= 64 and (
                blend == "deferred" or have or need <= _cache_budget_bytes(dev))
            if deferred:
                cache, busy = _cache_workspace(dev, cshape, cdtype)
                if busy is not None:
                    main.wait_event(busy)       # an earlier call's blend (any stream) still owns it
            else:
                acc = torch.zeros((1, orig[0], orig[1], orig[2], K), dtype=torch.float32, device=dev)
                cnt = torch.zeros((orig[0], orig[1], orig[2]), dtype=torch.float32, device=dev)

        # this build's own network says up front what it writes into the cache (``Net.cache_spec``): the first
        # window group then goes straight into the cache like the others (it used to run through the generic
        # path -- gather, forward, a 1 GB copy of its predictions into the cache -- to learn the class count)
        spec = getattr(owner, "cache_spec", None)
        spec = spec() if (callable(spec) and into is not None) else None
        if spec is not None:
            pick_strategy(*spec)

        for gi, g0 in enumerate(range(lo, hi, sw_batch_size)):
            grp = wins_u[g0:min(g0 + sw_batch_size, hi)]
            slot = g0 - lo
            # the first forward after a weight change builds the eval packs (folded BatchNorm, merged pairs) on the
            # stream it runs on: it stays on the caller's stream, in front of the fork, so that no lane can read a
            # pack that is still being written (round 4: with the cache allocated up front group 0 had moved to
            # lane 0 and a z-slab parity test failed once in a while)
            cold = False
            if lanes is not None and gi == 0 and callable(getattr(owner, "eval_state", None)):
                state = owner.eval_state()
                cold = getattr(owner, "_sw_warm_state", None) != state
                try:
                    owner._sw_warm_state = state          # (kept on the network object: dies with it)
                except AttributeError:
                    cold = True
            if cache is not None and into is not None and lanes is not None and not cold:
                # window groups are independent once their predictions go to the cache: alternate
                # them over two streams (each with its own activation lane of the engine) so the
                # small deep layers of one group overlap the wide layers of the other
                if not forked:
                    for st in lanes:
                        st.wait_stream(main)
                    forked = True
                lane = gi % len(lanes)
                with torch.cuda.stream(lanes[lane]):
                    if stats is not None:
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                    if volc is not None:
                        ok = into(ops.WindowBatch(volc, grp, roi), cache[slot:slot + len(grp)], lane)
                    else:
                        wbuf = torch.empty((len(grp), roi[0], roi[1], roi[2], Cin), dtype=window_dtype,
                                           device=dev)
                        ops.sw_gather(img, b, grp, wbuf)
                        ok = into(wbuf.permute(0, 4, 1, 2, 3), cache[slot:slot + len(grp)], lane)
                    if stats is not None:
                        e1 = torch.cuda.Event(enable_timing=True)
                        e1.record()
                        lane_events[lane].append((e0, e1))
                if ok:
                    blend_finished_levels(g0 + len(grp) - 1)
                    continue
                if gi > 1:
                    raise RuntimeError("predictor stopped accepting forward_into mid-volume")
                into = None          # e.g. a class count the cache layout cannot take directly
            if stats is not None and lanes is None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            if cache is not None and into is not None and volc is not None and \
                    into(ops.WindowBatch(volc, grp, roi), cache[slot:slot + len(grp)]):
                if stats is not None and lanes is None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    lane_events[0].append((e0, e1))
                blend_finished_levels(g0 + len(grp) - 1)
                continue                                   # windows read in place, predicted straight into the cache
            wbuf = torch.empty((len(grp), roi[0], roi[1], roi[2], Cin), dtype=window_dtype,
                               device=dev)
            ops.sw_gather(img, b, grp, wbuf)
            wview = wbuf.permute(0, 4, 1, 2, 3)
            if cache is not None and into is not None and into(wview, cache[slot:slot + len(grp)]):
                if stats is not None and lanes is None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    lane_events[0].append((e0, e1))
                blend_finished_levels(g0 + len(grp) - 1)
                continue                                   # predicted straight into the cache
            pn = as_ndhwc(predictor(wview))
            if K is None:                                  # first group: pick the strategy
                pick_strategy(pn.shape[4], pn.dtype)
            if deferred:
                cache[slot:slot + len(grp)].copy_(pn)
                blend_finished_levels(g0 + len(grp) - 1)
            else:
                ops.sw_scatter_add(pn, grp, acc, cnt, imp)
        if forked:
            for st in lanes:
                main.wait_stream(st)
        if stats is not None:
            stats["host_enqueue_s"] = stats.get("host_enqueue_s", 0.0) + (_time.perf_counter() - t_enq)
            stats.setdefault("host_enqueue_each_s", []).append(_time.perf_counter() - t_enq)
            stats["lanes"] = len(lanes) if lanes else 1
            stats.setdefault("lane_events", [[] for _ in lane_events])
            for dst, src in zip(stats["lane_events"], lane_events):
                dst.extend(src)
            t_enq = _time.perf_counter()
            lane_events = [[] for _ in lane_events]
        lab = None
        if cnt_p is not None:
            # slabs were blended on the way; whatever is left (nothing, when the last level ended the loop)
            blend_finished_levels(hi - 1)
            assert z_done == orig[0], (z_done, orig[0])
            main.wait_stream(_blend_stream(dev))
            acc, cnt, lab = acc_p, cnt_p, lab_p
            _cache_release_point(dev, main)
            cache = None
            outs.append(acc); cnts.append(cnt); labs.append(lab if return_labels else None)
            continue
        if not partial and return_labels:
            lab = torch.empty((out_d, orig[1], orig[2]),
                              dtype=torch.uint8 if K <= 256 else torch.int32, device=dev)
        if deferred:
            if want_logits:
                acc = torch.empty((1, out_d, orig[1], orig[2], K), dtype=torch.float32, device=dev)
            cnt = torch.empty((out_d, orig[1], orig[2]), dtype=torch.float32, device=dev)
            ops.sw_blend(cache, per_dim_b, lo, hi, roi, out_d, orig[1], orig[2], importance=imp,
                         out_logits=acc if want_logits else None, out_count=cnt, labels=lab,
                         normalize=not partial)
            _cache_release_point(dev, main)
            cache = None
        elif not partial:
            if lab is None:
                lab = torch.empty((orig[0], orig[1], orig[2]),
                                  dtype=torch.uint8 if K <= 256 else torch.int32, device=dev)
            ops.sw_finalize(acc, cnt, lab, write_logits=True)
        outs.append(acc); cnts.append(cnt); labs.append(lab if return_labels and not partial else None)
    logits = None
    if outs[0] is not None:
        logits = (torch.cat(outs, 0) if B > 1 else outs[0]).permute(0, 4, 1, 2, 3)
    if not return_labels:
        return logits
    # (one volume: views, not copies -- the count map of a 512^3 volume is 537 MB)
    one = B == 1
    labels = None if labs[0] is None else (labs[0][None] if one else torch.stack(labs)).unsqueeze(1)
    return SlidingWindowResult(logits, labels, cnts[0][None] if one else torch.stack(cnts))


def z_slabs(depth: int, world: int) -> List[Tuple[int, int]]:
    """`world` contiguous plane ranges covering [0, depth), sizes differing by at most one."""
    base, rem = divmod(int(depth), int(world))
    out, z = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((z, z + n))
        z += n
    return out


def gather_label_slabs(labels_slab: torch.Tensor, depth: int, rank: int, world: int, group=None):
    """All-gather the per-rank label slabs ([..., d_r, H, W], the slab of ``z_slabs(depth, world)[rank]``)
    into the full label volume on every rank.  Labels are 1 byte per voxel: 128 MB for a 512^3
    volume, against 8.6 GB for the f32 logits -- the only data-path exchange of sharded inference."""
    import torch.distributed as dist
    slabs = z_slabs(depth, world)
    dmax = max(b - a for a, b in slabs)
    lead = labels_slab.shape[:-3]
    pad = torch.zeros(lead + (dmax,) + tuple(labels_slab.shape[-2:]), dtype=labels_slab.dtype,
                      device=labels_slab.device)
    pad[..., :labels_slab.shape[-3], :, :] = labels_slab
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[..., :b - a, :, :] for p, (a, b) in zip(parts, slabs)], dim=-3)


class SlidingWindowInferer:
    """``monai.inferers.SlidingWindowInferer`` surface (roi_size, sw_batch_size, overlap, mode)."""

    def __init__(self, roi_size, sw_batch_size: int = 1, overlap: float = 0.25,
                 mode: str = "constant", sigma_scale: float = 0.125, device=None, **_unused):
        self.roi_size, self.sw_batch_size = roi_size, sw_batch_size
        self.overlap, self.mode, self.sigma_scale, self.device = overlap, mode, sigma_scale, device

    def __call__(self, inputs: torch.Tensor, network: Callable, return_labels: bool = False):
        return sliding_window_inference(inputs, self.roi_size, self.sw_batch_size, network,
                                        self.overlap, self.mode, self.sigma_scale, self.device,
                                        return_labels=return_labels)
