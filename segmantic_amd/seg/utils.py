"""Device selection, as reference ``src/segmantic/seg/utils.py:4-12``."""
import torch


def make_device(gpu_ids):
    """[] + cuda available -> cuda:0 ; [] / negative first id -> cpu ; else cuda:{gpu_ids[0]}."""
    if not gpu_ids and torch.cuda.is_available():
        gpu_ids = [0]
    if not gpu_ids or gpu_ids[0] < 0:
        return torch.device("cpu")
    return torch.device(f"cuda:{gpu_ids[0]}")
