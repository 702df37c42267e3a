#!/bin/bash
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" > gpurun_out/r4/j_tests.log 2>&1 || { tail -30 gpurun_out/r4/j_tests.log; exit 1; }
tail -2 gpurun_out/r4/j_tests.log
bash scripts/gpu_kstats_c4.sh 8 > gpurun_out/r4/kstats_c4_b8.log 2>&1 || { tail gpurun_out/r4/kstats_c4_b8.log; exit 1; }
cp gpurun_out/r4/kstats_c4.txt gpurun_out/r4/kstats_c4_b8.txt
bash scripts/gpu_kstats_c4.sh 4 > gpurun_out/r4/kstats_c4_b4.log 2>&1 || exit 1
cp gpurun_out/r4/kstats_c4.txt gpurun_out/r4/kstats_c4_b4.txt
bash scripts/gpu_kstats.sh train 10 > gpurun_out/r4/e_kstats.log 2>&1 || exit 1
f=$(ls -t gpurun_out/prof/ks_train/*/*kernel_trace.csv | head -1)
python3 scripts/step_timeline.py $f -v > gpurun_out/r4/timeline_train.txt
head -8 gpurun_out/r4/timeline_train.txt
