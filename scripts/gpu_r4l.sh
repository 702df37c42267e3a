#!/bin/bash
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "pair or split_act or wpack or fin or stats" > gpurun_out/r4/l_tests.log 2>&1 || { tail -40 gpurun_out/r4/l_tests.log; exit 1; }
tail -2 gpurun_out/r4/l_tests.log
timeout -k 10 900 python -m pytest tests/test_unet_gpu.py tests/test_e2e_gpu.py -m gpu -x -q > gpurun_out/r4/l_tests2.log 2>&1 || { tail -40 gpurun_out/r4/l_tests2.log; exit 1; }
tail -2 gpurun_out/r4/l_tests2.log
bash scripts/gpu_sweep.sh "SEGMI_PAIR_TRAIN=0" "SEGMI_PAIR_TRAIN=1" || exit 1
