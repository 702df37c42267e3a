"""The package's side streams, ONE set per device.

Every engine, the sliding-window driver, the samplers and the gradient exchange used to create streams of their
own.  A process that builds several networks (``bench.py``: a training net, an inference net, an f32 net, a fit net
...) then holds a dozen HIP streams over the runtime's 4 hardware queues, and which of them alias one queue -- i.e.
run in order -- depends on the creation history: the three inference lanes overlapped when the inference leg ran in
a fresh process (lane busy times 36 / 34 / 30 ms of a 39 ms volume, 25.6 volumes/s) and mostly serialised after a
training leg in the same process (18 / 20 / 23 ms, 24.0-24.7 volumes/s; round 4, `gpurun_out/r4/i_*.json`).

Sharing is safe: a stream only orders the work put on it; every dependency in this package is an explicit event
wait, so two users of one stream gain ordering between their launches, never lose any.  Slots:

    0  weight gradients (training)      | inference lane 0
    1  weight re-pack (training)        | inference lane 1
    2  residual branch / batch sampler  | inference lane 2
    3  gradient buckets (data-parallel) | ordered blend of the sliding-window driver
    4+ further inference lanes
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

WGRAD, PACK, AUX, EXCHANGE = 0, 1, 2, 3
BLEND = EXCHANGE

_SHARED: Dict[Tuple[int, int], "torch.cuda.Stream"] = {}


def shared_stream(device, slot: int) -> "torch.cuda.Stream":
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, int(slot))
    st = _SHARED.get(key)
    if st is None:
        # create the lower slots first: the slot -> stream (-> hardware queue) order is the same in every process
        for s in range(int(slot) + 1):
            if (idx, s) not in _SHARED:
                _SHARED[(idx, s)] = torch.cuda.Stream(device=torch.device("cuda", idx))
        st = _SHARED[key]
    return st
