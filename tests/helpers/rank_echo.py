"""Rank program for tests/test_launch.py: joins the gloo group through the package's own
``init_distributed``, all-reduces its rank, rank 0 prints one JSON line (the bench.py contract).
``--fail-rank R`` makes rank R exit with code 3 after the collective."""
import json
import os
import sys

import torch
import torch.distributed as dist

from segmantic_amd.seg.distributed import init_distributed


def main():
    fail = int(sys.argv[sys.argv.index("--fail-rank") + 1]) if "--fail-rank" in sys.argv else -1
    rank, local_rank, world = init_distributed(backend="gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"world": world, "sum": float(t.item()), "hwq": os.environ.get("GPU_MAX_HW_QUEUES"),
                          "argv": sys.argv[1:]}), flush=True)
    dist.destroy_process_group()
    if rank == fail:
        sys.exit(3)


if __name__ == "__main__":
    main()
