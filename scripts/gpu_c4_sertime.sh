#!/bin/bash
# serial per-kernel times of the C4-shaped step (B=2 x 160^3, K=32) on one GPU
set -o pipefail
mkdir -p gpurun_out/r3 gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/c4_ser
SEGMI_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/c4_ser -- python3 bench.py --workload train --size 160 --classes 32 --batch 2 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/prof/c4_ser.log 2>&1 || exit 1
python scripts/grid_table.py $(ls -t gpurun_out/prof/c4_ser/*/*kernel_trace.csv | head -1) > gpurun_out/r3/c4_ser_by_grid.txt
tail -1 gpurun_out/r3/c4_ser_by_grid.txt
