"""``segmantic.seg.enum`` (reference ``src/segmantic/seg/enum.py``)."""
from enum import Enum


class EnsembleCombination(str, Enum):
    mean = "mean"
    vote = "vote"
    select_best = "select_best"
