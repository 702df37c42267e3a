#!/bin/bash
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > gpurun_out/r4/e_tests.log 2>&1 || { tail -30 gpurun_out/r4/e_tests.log; exit 1; }
tail -2 gpurun_out/r4/e_tests.log
bash scripts/gpu_kstats.sh train 10 > gpurun_out/r4/e_kstats.log 2>&1 || exit 1
f=$(ls -t gpurun_out/prof/ks_train/*/*kernel_trace.csv | head -1)
python3 scripts/step_timeline.py $f -v > gpurun_out/r4/timeline_train.txt
head -12 gpurun_out/r4/timeline_train.txt
