#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 300 python -m pytest tests/test_pipeline.py -m gpu -q -x 2>&1 | grep -E "Error|error|assert|passed|failed" | head -20
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r2/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2/trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r2/trace.log 2>&1 || { tail -5 gpurun_out/r2/trace.log; exit 1; }
f=$(find gpurun_out/r2/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/timeline.py $f > gpurun_out/r2/timeline.txt 2>&1; head -60 gpurun_out/r2/timeline.txt
