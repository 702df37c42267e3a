#!/bin/bash
# serial per-kernel times of the training step (one stream, SEGMI_SERIAL=1) + the fused-BN micro-benchmark
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 120 python scripts/fused_bn_bench.py 2>&1 | tee gpurun_out/r3/fused_bn_bench.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/ser_time; mkdir -p gpurun_out/prof
SEGMI_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/ser_time -- python3 bench.py --workload train --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/prof/ser_time.log 2>&1 || exit 1
python scripts/grid_table.py $(ls -t gpurun_out/prof/ser_time/*/*kernel_trace.csv | head -1) > gpurun_out/r3/ser_by_grid.txt
tail -1 gpurun_out/r3/ser_by_grid.txt
