#!/bin/bash
# round-2 quick check: GPU tests, the two dominant-kernel micro timings, the default bench line
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/r2/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python scripts/ring2_diag.py > gpurun_out/r2/ring2.log 2>&1 && cat gpurun_out/r2/ring2.log || exit 1
timeout -k 10 120 python scripts/wgrad_diag.py > gpurun_out/r2/wgrad.log 2>&1 && cat gpurun_out/r2/wgrad.log || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err && cat gpurun_out/r2/bench.json || exit 1
