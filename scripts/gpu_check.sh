#!/bin/bash
# full GPU test suite + the default bench line (one gpurun call); logs under gpurun_out/r4
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gputests.log 2>&1; rc=$?
tail -5 gpurun_out/r4/gputests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err || exit 1
tail -c 3000 gpurun_out/r4/bench_default.json
